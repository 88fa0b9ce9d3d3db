// k_lists.hip -- K3 / K4 / K5 of the rotated variants on LISTS: the part of a run behind the region query touches only
// the PETs that can matter.
//
// The reference expands clusters from core points only (cLoops/cDBSCAN2.py:114-192 queryGrid, :304-346
// getSparseCellNeighbor; cLoops/cDBSCAN.py:155-184 expandCluster).  Rounds 1-4 walked LDS tiles of EVERY PET of the run for
// that (chains, cross-strip unions, border rule, labels): 35-60 % of a run's PETs are cores, 20-25 % are non-core PETs with
// a neighbour, the rest is isolated noise that was staged, searched past and stepped over again and again.  Here the run
// is split once, right behind K2:
//
//   k_classify    one pass over the run's (q, sp, K2 word): one bit per PET "core", one bit "walker" (a non-core PET with a
//                 neighbour: the only PETs a border rule can label), their counts per 64-PET group, exclusive inside a
//                 2048-PET tile; the tiles' offsets by the last workgroup to finish (no launch of its own, nothing spins).
//                 Variant 2 also needs the smallest input row of every rotated cell (cDBSCAN2.py:117: dict insertion order)
//                 -- a minimum over ALL PETs of the cell, so it is taken here, by LDS atomics on the tile's own cells.
//   k_make_lists  second pass: the cores (q, sp, position, key) and the walkers (q, sp, position, K2 word) written to compact
//                 arrays in sorted order; the rank index cgrank[g] = cores in front of 64-PET group g (with the core bit mask:
//                 any POSITION of the layout -> index into the core array in two loads, so K2's window hints stay usable);
//                 the cores-only strip table.
//   k_chain_c     cores of one strip within eps in q form a chain = a contiguous run of the core array: chain head per core
//                 (a ballot inside the wave, one look-back per wave), the chain's last core, per-head resets.
//   k_union_c     cross-strip core-core edges by lock-free union-find on the chain heads; tiles of the CORE array with a
//                 left halo (the window one strip below holds cores only: nothing to step over).
//   k_flatten_c   root per core, component key / core count per root, the root list.
//   k_border_w    the border rule (R1 / R2) for the walkers; the cores around a tile's position range are
//                 a contiguous slice of the core array, staged once; a walker's windows start at rank(K2 hint).
//   k_emit_records_w, k_final_lists: the release records of variant 2 and labels / cluster table / distance list from the
//                 two lists.
//
// Results are bit-identical to the tile kernels of cloops_hip.hip (cl_set_traversal switches between them; tests compare).
#include "cl_chrom.h"

#define LT 2048                  // positions per tile of k_classify / k_make_lists (256 threads x 8)
#define LG (LT / 64)             // 64-PET groups per tile
#define L_HALO 128               // variant 2: staged halo of k_classify (cell heads look one PET back, cells run on behind the tile)

// XCD-aware tile order (workgroup b runs on XCD b % 8: runs of consecutive tiles share halos in that XCD's L2)
#define L_RUN 8
__device__ __forceinline__ int ltile_of_block(int bid)
{
    const int xcd = bid & 7, kseq = bid >> 3;
    return ((kseq / L_RUN) * 8 + xcd) * L_RUN + (kseq % L_RUN);
}
static inline int ltile_grid(int ntiles) { return ((ntiles + 8 * L_RUN - 1) / (8 * L_RUN)) * (8 * L_RUN); }

__device__ __forceinline__ unsigned long long low_mask(int lane) { return (1ull << lane) - 1ull; }      // bits below `lane` (0 .. 63)
__device__ __forceinline__ int lane_rank(unsigned long long m)      // set bits of m in front of the calling lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// number of cores in front of position p of the run's layout (p may be M: all of them)
__device__ __forceinline__ int core_rank(const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank, int p)
{
    const int gi = p >> 6;
    return cgrank[gi] + __popcll(cmask[gi] & low_mask(p & 63));
}

// ------------------------------------------------------------------------------------------
// k_classify
// ------------------------------------------------------------------------------------------
template <bool V2>
__global__ void __launch_bounds__(256)
k_classify(GridParams g, const int* __restrict__ sv, const int* __restrict__ sa, const int* __restrict__ strip_start, WordSrc ws,
           const u32* __restrict__ srow, unsigned long long* __restrict__ cmask, unsigned long long* __restrict__ wmask,
           unsigned long long* __restrict__ hmask, int* __restrict__ cgloc, int* __restrict__ wgloc,
           int* __restrict__ bsum /* [2][nblk + 1] */, int* __restrict__ boff /* [2][nblk + 1] */, int* ticket, int* __restrict__ lcnt,
           int* __restrict__ cellfirst)
{
    constexpr int WIN = LT + 2 * L_HALO;
    __shared__ __attribute__((aligned(16))) int2 lw[V2 ? WIN : 1];
    __shared__ int lmin[V2 ? LT : 1];
    __shared__ unsigned long long l_chead[LG];
    __shared__ int l_cc[LG], l_wc[LG];
    __shared__ int l_hlast, l_is_last;
    __shared__ int red[4];
    const int M = strip_start[g.S];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT, base = t0 - L_HALO;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nmask = ~(g.peps - 1);
    int q[8], sp[8], w[8];
    u32 row[8];
    if (V2) {
        // the tile plus a halo as (q, sp) pairs: the padded sorted arrays take unpredicated 16-byte loads
        const int4* __restrict__ gq4 = reinterpret_cast<const int4*>(sv + base);
        const int4* __restrict__ gp4 = reinterpret_cast<const int4*>(sa + base);
        int4* l4 = reinterpret_cast<int4*>(lw);
        for (int cidx = threadIdx.x; cidx < WIN / 4; cidx += 256) {
            const int4 a = gq4[cidx], b = gp4[cidx];
            l4[2 * cidx] = make_int4(a.x, b.x, a.y, b.y);
            l4[2 * cidx + 1] = make_int4(a.z, b.z, a.w, b.w);
        }
        for (int k = threadIdx.x; k < LT; k += 256) lmin[k] = INT_MAX;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = t0 + u * 256 + (int)threadIdx.x; row[u] = i < M ? srow[i] : 0u; }
        if (threadIdx.x == 0) l_hlast = -1;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int2 me = lw[L_HALO + u * 256 + (int)threadIdx.x]; q[u] = me.x; sp[u] = me.y; }
    } else {
        const bool need = ws.D != nullptr;               // (a run on the layout its words were made on finds them by position alone)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            q[u] = (need && i < M) ? sv[i] : 0; sp[u] = (need && i < M) ? sa[i] : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = t0 + u * 256 + (int)threadIdx.x; w[u] = i < M ? ws.raw(i, q[u], sp[u]) : 0; }
    unsigned headbits = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < M;
        const bool core = in && cw_core(w[u], g.minPts);
        const bool walk = in && !core && cw_count(w[u]) > 1;      // (count <= 1: nothing within eps -- most of the background noise ends here)
        const unsigned long long cb = __ballot(core), wb = __ballot(walk);
        const int k = u * 4 + wv, gidx = (t0 >> 6) + k;
        if (lane == 0) { cmask[gidx] = cb; wmask[gidx] = wb; l_cc[k] = __popcll(cb); l_wc[k] = __popcll(wb); }
        if (V2) {
            // a PET starts a rotated cell (strip, q / eps) iff its predecessor lies in an earlier strip or below the cell's lower q
            // edge (cDBSCAN2.py:67-70; variant 2 runs with A0 = V0 = 0)
            const int2 pv = lw[L_HALO + u * 256 + (int)threadIdx.x - 1];
            const int q0 = div_eps(g, q[u]) * g.eps;
            const bool start = in && (i == 0 || (pv.y & nmask) != (sp[u] & nmask) || pv.x < q0);
            const unsigned long long hb = __ballot(start);
            if (lane == 0) { l_chead[k] = hb; hmask[gidx] = hb; }
        }
    }
    if (blk == nblk - 1 && threadIdx.x == 0) {
        // one group behind the last tile: position M may be its first (rank(M) = every core)
        const int ge = nblk * LG;
        cmask[ge] = 0ull; wmask[ge] = 0ull; if (V2) hmask[ge] = 0ull;
        cgloc[ge] = 0; wgloc[ge] = 0;
    }
    __syncthreads();
    int totC = 0, totW = 0;
    if (threadIdx.x < 64) {
        // exclusive counts of the tile's 32 groups: lanes 0..31 the cores, 32..63 the walkers
        const int k = lane & 31;
        const int v = lane < 32 ? l_cc[k] : l_wc[k];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up(incl, d, 32); incl += k >= d ? t : 0; }
        (lane < 32 ? cgloc : wgloc)[(t0 >> 6) + k] = incl - v;
        totC = __shfl(incl, 31); totW = __shfl(incl, 63);
    }
    if (V2) {
        // cellfirst[cell head] = the smallest input row of the cell's PETs, by LDS atomics on the cells that begin in this tile
        // (a cell that began in an earlier tile: that tile walks it, below)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            if (i >= M) continue;
            const int k = u * 4 + wv;
            const unsigned long long upto = l_chead[k] & ((2ull << lane) - 1ull);
            int pos = -1;
            if (upto) pos = (i - lane) + 63 - __clzll((long long)upto);
            else
                for (int k2 = k - 1; k2 >= 0; --k2) {
                    const unsigned long long o2 = l_chead[k2];
                    if (o2) { pos = t0 + 64 * k2 + 63 - __clzll((long long)o2); break; }
                }
            if (pos >= t0) atomicMin(&lmin[pos - t0], (int)row[u]);
            headbits |= (pos == i ? 1u : 0u) << u;
            if (i == min(t0 + LT, M) - 1) l_hlast = pos;
        }
        __syncthreads();
        const int tend = t0 + LT;
        if (threadIdx.x < 64 && tend < M && l_hlast >= t0) {
            // the cell of the tile's last PET may go on behind the tile: wave 0 walks it, 64 PETs per round (right halo, then global memory)
            const int2 lp = lw[L_HALO + LT - 1];
            const int p0 = lp.y & nmask, qend = div_eps(g, lp.x) * g.eps + g.eps;
            int m = INT_MAX;
            for (int j0 = tend; j0 < M; j0 += 64) {
                const int j = j0 + (int)threadIdx.x;
                bool in = j < M;
                if (in) {
                    const int2 cc = j < base + WIN ? lw[j - base] : make_int2(sv[j], sa[j]);
                    in = (cc.y & nmask) == p0 && cc.x < qend;
                }
                if (in) m = min(m, (int)srow[j]);
                if (__ballot(in) != ~0ull) break;        // the cell ends inside this round (its PETs are contiguous)
            }
            m = dpp_reduce_wave(m, OpMin());
            if (threadIdx.x == 0 && m != INT_MAX) atomicMin(&lmin[l_hlast - t0], m);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (headbits & (1u << u)) { const int i = t0 + u * 256 + (int)threadIdx.x; cellfirst[i] = lmin[i - t0]; }
    }
    // the tiles' offsets: both block sums go out as device-scope atomics whose return is awaited before the ticket is taken
    // (cl_common.h scan_tail_last_block: a release fence would write back the XCD's whole L2); the last workgroup scans them
    if (threadIdx.x == 0) {
        const int o1 = __hip_atomic_exchange(&bsum[blk], totC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int o2 = __hip_atomic_exchange(&bsum[nblk + 1 + blk], totW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" :: "v"(o1), "v"(o2) : "memory");
        l_is_last = atomicAdd(ticket, 1) == nblk - 1;
    }
    __syncthreads();
    if (!l_is_last) return;
    int carryC = 0, carryW = 0;
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int k = b0 + (int)threadIdx.x;
        const int vc = k < nblk ? __hip_atomic_load(&bsum[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int vw = k < nblk ? __hip_atomic_load(&bsum[nblk + 1 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int tc, tw;
        const int ic = wg256_inclusive_scan(vc, red, tc);
        const int iw = wg256_inclusive_scan(vw, red, tw);
        if (k < nblk) { boff[k] = carryC + ic - vc; boff[nblk + 1 + k] = carryW + iw - vw; }
        carryC += tc; carryW += tw;
    }
    if (threadIdx.x == 0) {
        boff[nblk] = carryC; boff[2 * nblk + 1] = carryW;
        lcnt[0] = carryC; lcnt[1] = carryW;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------
// k_make_lists
// ------------------------------------------------------------------------------------------
template <bool V2>
__global__ void __launch_bounds__(256)
k_make_lists(GridParams g, const int* __restrict__ sv, const int* __restrict__ sa, const int* __restrict__ strip_start, WordSrc ws,
             const u32* __restrict__ srow, const unsigned long long* __restrict__ cmask, const unsigned long long* __restrict__ wmask,
             const unsigned long long* __restrict__ hmask, const int* __restrict__ cgloc, const int* __restrict__ wgloc,
             const int* __restrict__ boff, const int* __restrict__ cellfirst, int* __restrict__ cgrank, int* __restrict__ wgrank,
             int2* __restrict__ cpair, int* __restrict__ cpos, int* __restrict__ ckey, int2* __restrict__ wpair,
             int* __restrict__ wpos, int* __restrict__ wenc, int* __restrict__ cstrip)
{
    const int M = strip_start[g.S];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cbase = boff[blk], wbase = boff[nblk + 1 + blk];
    const int C = boff[nblk], W = boff[2 * nblk + 1];
    // the cores-only strip table: cstrip[s] = cores in front of strip s (cstrip[S] = cstrip[S + 1] = C)
    for (int u = blk * 256 + (int)threadIdx.x; u <= g.S + 1; u += nblk * 256) {
        const int p = u <= g.S ? strip_start[u] : M;
        int r = C;
        if (p < M) { const int gi = p >> 6; r = boff[p / LT] + cgloc[gi] + __popcll(cmask[gi] & low_mask(p & 63)); }
        cstrip[u] = r;
    }
    if (blk == nblk - 1 && threadIdx.x == 0) { cgrank[nblk * LG] = C; wgrank[nblk * LG] = W; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const int k = u * 4 + wv, gidx = (t0 >> 6) + k;
        const unsigned long long cb = cmask[gidx], wb = wmask[gidx];          // (wave-uniform)
        const int cg = cbase + cgloc[gidx], wg = wbase + wgloc[gidx];
        if (lane == 0) { cgrank[gidx] = cg; wgrank[gidx] = wg; }
        const bool isc = (cb >> lane) & 1ull, isw = (wb >> lane) & 1ull;
        if (!(isc | isw)) continue;
        const int q = sv[i], sp = sa[i];
        if (isc) {
            const int dst = cg + __popcll(cb & low_mask(lane));
            int key;
            if (V2) {
                // the component key is a minimum over the CELLS of its cores (cDBSCAN2.py:117-140): of a cell's cores only the
                // first carries the cell (two cores of one cell are always one component).  The cell's head = the latest
                // cell-opening PET at or before the core; "first core" = no core between the head and it.
                const unsigned long long hm = hmask[gidx];
                const unsigned long long upto = hm & ((2ull << lane) - 1ull);
                int head;
                bool first;
                if (upto) {
                    const int hb = 63 - __clzll((long long)upto);
                    head = (i - lane) + hb;
                    first = (cb & low_mask(lane) & ~low_mask(hb)) == 0ull;
                } else {
                    first = (cb & low_mask(lane)) == 0ull;
                    head = 0;
                    for (int g2 = gidx - 1; g2 >= 0; --g2) {
                        const unsigned long long h2 = hmask[g2], c2 = cmask[g2];
                        if (h2) {
                            const int hb = 63 - __clzll((long long)h2);
                            head = g2 * 64 + hb;
                            if (c2 >> hb) first = false;
                            break;
                        }
                        if (c2) first = false;
                    }
                }
                key = first ? cellfirst[head] : INT_MAX;
            } else {
                key = (int)srow[i];                       // variant 1: the component's start point is its smallest-row core (cDBSCAN.py:134-137)
            }
            cpair[dst] = make_int2(q, sp); cpos[dst] = i; ckey[dst] = key;
        } else {
            const int dst = wg + __popcll(wb & low_mask(lane));
            wpair[dst] = make_int2(q, sp); wpos[dst] = i; wenc[dst] = ws.word(i, q, sp);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_chain_c: chains on the core array
// ------------------------------------------------------------------------------------------
// Inside a strip every pair is within eps in the strip coordinate, so cores whose q gaps are <= eps form a CHAIN -- on the core
// array a contiguous run.  chainid[c] = index of the chain's first core (its head: the union-find node of all its cores);
// cend[head] = index of its last core.  Every component root is a chain head: the per-root accumulators are reset here.
#define CH_PER 4
__global__ void __launch_bounds__(256)
k_chain_c(GridParams g, const int* __restrict__ lcnt, const int2* __restrict__ cpair, int* __restrict__ chainid,
          int* __restrict__ parent, int* __restrict__ compkey, int* __restrict__ ncore, int* __restrict__ bsize,
          int* __restrict__ usize, int* __restrict__ state, int* __restrict__ cend)
{
    const int C = lcnt[0];
    const int lane = threadIdx.x & 63;
    const int nmask = ~(g.peps - 1);
    int2 me[CH_PER], pv[CH_PER], nx[CH_PER];
#pragma unroll
    for (int e = 0; e < CH_PER; ++e) {
        const int c = (blockIdx.x * CH_PER + e) * 256 + (int)threadIdx.x;
        const bool in = c < C;
        me[e] = in ? cpair[c] : make_int2(0, 0);
        pv[e] = (in && c > 0) ? cpair[c - 1] : make_int2(0, INT_MIN);
        nx[e] = (in && c + 1 < C) ? cpair[c + 1] : make_int2(0, INT_MIN);
    }
#pragma unroll
    for (int e = 0; e < CH_PER; ++e) {
        const int c = (blockIdx.x * CH_PER + e) * 256 + (int)threadIdx.x;
        const bool in = c < C;
        const int p0 = me[e].y & nmask;
        const bool open = in && (c == 0 || (pv[e].y & nmask) != p0 || pv[e].x < me[e].x - g.eps);
        const bool last = in && (c + 1 >= C || (nx[e].y & nmask) != p0 || nx[e].x > me[e].x + g.eps);
        const unsigned long long ob = __ballot(open);
        const unsigned long long upto = ob & ((2ull << lane) - 1ull);
        int head = upto ? (c - lane) + 63 - __clzll((long long)upto) : -1;
        if (__any(in && !upto)) {
            // the chain of the wave's first cores opened in front of the wave: one look-back for all of them, 64 cores per round
            // (core 0 opens a chain: the loop ends)
            int found = -1;
            for (int k0 = (c - lane) - 64; found < 0 && k0 >= 0; k0 -= 64) {
                const int j = k0 + lane;
                const int2 a = cpair[j];
                const int2 b = j > 0 ? cpair[j - 1] : make_int2(0, INT_MIN);
                const bool o = j == 0 || (b.y & nmask) != (a.y & nmask) || b.x < a.x - g.eps;
                const unsigned long long bal = __ballot(o);
                if (bal) found = k0 + 63 - __clzll((long long)bal);
            }
            if (!upto) head = found;
        }
        if (in) {
            chainid[c] = head;
            if (head == c) { parent[c] = c; compkey[c] = INT_MAX; ncore[c] = 0; bsize[c] = 0; usize[c] = 0; state[c] = ST_LIVE; }
            if (last) cend[head] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_union_c: cross-strip core-core edges
// ------------------------------------------------------------------------------------------
// A core i of strip s against the cores of strip s-1 in its window (the pairs with strip s+1 are handled from the other
// endpoint).  Chains, not points, are what has to be united: every lane collects the distinct chains B of strip s-1 it touches,
// the wave then keeps ONE lane per distinct (own chain A, chain B) pair, and only those lanes run the union-find step.
#define LU_MAXB 4
__device__ __forceinline__ int lower_bound_pairs(const int2* __restrict__ pv, int lo, int hi, int val)      // first j of [lo, hi) with pv[j].x >= val
{
    while (hi - lo > 4) {
        const int qq = (hi - lo) >> 2;
        const int m1 = lo + qq, m2 = m1 + qq, m3 = m2 + qq;
        const int v1 = pv[m1].x, v2 = pv[m2].x, v3 = pv[m3].x;
        if (v1 >= val) hi = m1;
        else if (v2 >= val) { lo = m1 + 1; hi = m2; }
        else if (v3 >= val) { lo = m2 + 1; hi = m3; }
        else lo = m3 + 1;
    }
    while (lo < hi && pv[lo].x < val) ++lo;
    return lo;
}

template <int NT, int HALO>
__global__ void __launch_bounds__(256)
k_union_c(GridParams g, int ntiles, const int* __restrict__ lcnt, const int2* __restrict__ cpair, const int* __restrict__ chainid,
          const int* __restrict__ cstrip, const int* __restrict__ cend, int* parent)
{
    constexpr int WIN = NT + HALO;
    __shared__ int2 lw[WIN];
    __shared__ int lx[WIN];
    const int C = lcnt[0];
    const int tile = ltile_of_block(blockIdx.x);
    const int t0 = tile * NT;
    if (tile >= ntiles || t0 >= C) return;
    const int base = t0 - HALO;
    for (int k = threadIdx.x; k < WIN; k += 256) {
        const int gi = base + k;
        const bool in = gi >= 0 && gi < C;
        lw[k] = in ? cpair[gi] : (gi < 0 ? make_int2(0, INT_MIN) : make_int2(INT_MAX, INT_MAX));
        lx[k] = in ? chainid[gi] : -1;
    }
    __syncthreads();
    const int wbeg = max(base, 0);
    LdsPairs w; w.a = lw; w.base = base;
    const int lane = threadIdx.x & 63;
#pragma unroll 1
    for (int u = 0; u < NT / 256; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < C;
        int Bs[LU_MAXB];
#pragma unroll
        for (int k = 0; k < LU_MAXB; ++k) Bs[k] = -1;
        int nb = 0;
        int A = -1;
        if (in) {
            const int2 me = lw[i - base];
            A = lx[i - base];
            const int s = me.y >> g.rbits;
            if (s > 0) {
                int tb = cstrip[s - 1];
                const int b = cstrip[s];
                const int qlo = me.x - g.eps, qhi = me.x + g.eps;
                const int T = me.y - g.peps;             // every candidate lies one strip below: "within eps in p" is sp_j >= sp_i - peps
                // strip s-1 ends where strip s begins, i.e. inside the staged range; if its first staged core lies below qlo the part
                // in front of the window cannot hold a candidate (sorted by q) and the staged part is the whole search range
                if (tb < wbeg && wbeg < b && lw[wbeg - base].x < qlo) tb = wbeg;
                auto touch = [&](int B) {
                    bool seen = false;
#pragma unroll
                    for (int k = 0; k < LU_MAXB; ++k) seen |= (Bs[k] == B);
                    if (seen) return;
                    if (nb < LU_MAXB) {
#pragma unroll
                        for (int k = 0; k < LU_MAXB; ++k) if (k == nb) Bs[k] = B;
                        ++nb;
                    } else {
                        uf_unite(parent, A, B);            // more chains than slots: unite right away
                    }
                };
                if (tb >= wbeg && b - tb <= 2047) {
                    // the window is staged.  Once a chain has been touched the walk jumps behind its last core (cend): a window
                    // covered by one chain costs one candidate instead of all of them.
                    const int len = b - tb;
                    int j;
                    if (len <= 31) j = lds_lower_bound8<5>(w, tb, b, qlo);
                    else if (len <= 63) j = lds_lower_bound8<6>(w, tb, b, qlo);
                    else if (len <= 255) j = lds_lower_bound8<8>(w, tb, b, qlo);
                    else j = lds_lower_bound8<11>(w, tb, b, qlo);
                    while (j < b) {
                        int2 cv[4]; int bv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { const int idx = min(j + k, b - 1); cv[k] = lw[idx - base]; bv[k] = lx[idx - base]; }
                        int next = j + 4;
                        bool stop = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (stop || j + k >= b) continue;
                            if (cv[k].x > qhi) { stop = true; next = b; continue; }
                            if (cv[k].y >= T) {
                                touch(bv[k]);
                                // (a short window is cheaper walked through: touch() passes over a chain it has seen)
                                if (len > 24) { stop = true; next = cend[bv[k]] + 1; }      // (> j + k: the chain holds this core)
                            }
                        }
                        j = next;
                    }
                } else {
                    // the strip below starts in front of the staged range (a strip population beyond the halo): global memory
                    int k = lower_bound_pairs(cpair, tb, b, qlo);
                    while (k < b) {
                        int2 cv[4]; int bv[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const int idx = min(k + e, b - 1); cv[e] = cpair[idx]; bv[e] = chainid[idx]; }
                        int next = k + 4;
                        bool stop = false;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (stop || k + e >= b) continue;
                            if (cv[e].x > qhi) { stop = true; next = b; continue; }
                            if (cv[e].y >= T) {
                                touch(bv[e]);
                                stop = true;
                                next = cend[bv[e]] + 1;
                            }
                        }
                        k = next;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < LU_MAXB; ++k) {
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
}

// ------------------------------------------------------------------------------------------
// k_flatten_c: root per core, component keys and core counts, the root list
// ------------------------------------------------------------------------------------------
//   variant 1: key = smallest input row of a core = the component's start point (cDBSCAN.py:134-137)
//   variant 2: key = smallest cellfirst over the cells holding its cores (cDBSCAN2.py:117-140)
// both arrive as ckey[c] (k_make_lists); two-level reduce-by-key as in k_flatten (cloops_hip.hip)
__global__ void __launch_bounds__(BIGTPB)
k_flatten_c(const int* __restrict__ lcnt, const int* __restrict__ chainid, const int* __restrict__ parent, const int* __restrict__ ckey,
            int* __restrict__ croot, int* __restrict__ compkey, int* __restrict__ ncore, int* __restrict__ rootlist,
            int* __restrict__ counters)
{
    __shared__ int hkey[AGG_H], hmin[AGG_H], hcnt[AGG_H];
    __shared__ int l_nroot, l_rootbase;
    if (threadIdx.x < AGG_H) { hkey[threadIdx.x] = -1; hmin[threadIdx.x] = INT_MAX; hcnt[threadIdx.x] = 0; }
    if (threadIdx.x == 0) l_nroot = 0;
    __syncthreads();
    const int C = lcnt[0];
    if ((int)blockIdx.x * BIGTPB * FLAT_PER >= C) return;
    int ii[FLAT_PER], r[FLAT_PER], key[FLAT_PER], x[FLAT_PER];
    bool in[FLAT_PER];
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        ii[e] = (blockIdx.x * FLAT_PER + e) * BIGTPB + (int)threadIdx.x;
        in[e] = ii[e] < C;
        x[e] = in[e] ? chainid[ii[e]] : -1;
        key[e] = in[e] ? ckey[ii[e]] : INT_MAX;
    }
    {
        // the union kernel has completed (kernel boundary = coherent): plain loads, all walks of the thread step together
        bool todo = false;
#pragma unroll
        for (int e = 0; e < FLAT_PER; ++e) todo |= in[e];
        while (todo) {
            int p[FLAT_PER];
#pragma unroll
            for (int e = 0; e < FLAT_PER; ++e) p[e] = in[e] ? parent[x[e]] : -1;
            todo = false;
#pragma unroll
            for (int e = 0; e < FLAT_PER; ++e) { todo |= in[e] && p[e] != x[e]; x[e] = in[e] ? p[e] : x[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        r[e] = in[e] ? x[e] : -1;
        if (in[e]) croot[ii[e]] = r[e];
    }
    const int lane = threadIdx.x & 63;
    int myslot[FLAT_PER];
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        myslot[e] = -1;
        const bool isroot = r[e] == ii[e] && r[e] >= 0;
        const unsigned long long rb = __ballot(isroot);
        if (rb) {
            int wbase = 0;
            if (lane == 0) wbase = atomicAdd(&l_nroot, __popcll(rb));
            wbase = __builtin_amdgcn_readfirstlane(wbase);
            if (isroot) myslot[e] = wbase + lane_rank(rb);
        }
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        const unsigned long long pending = __ballot(r[e] >= 0);
        if (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int R = __builtin_amdgcn_readlane(r[e], leader);
            const unsigned long long m = __ballot(r[e] == R);
            if (m == pending && __popcll(m) >= 16) {
                // a wave of one component (the inside of a large cluster): one reduction, one insertion by its first lane
                int mk = r[e] == R ? key[e] : INT_MAX;
                mk = dpp_reduce_wave(mk, OpMin());
                if (lane == leader) {
                    const int sl = agg_slot(hkey, R);
                    if (sl >= 0) { atomicMin(&hmin[sl], mk); atomicAdd(&hcnt[sl], __popcll(m)); }
                    else { atomicMin(&compkey[R], mk); atomicAdd(&ncore[R], __popcll(m)); }
                }
            } else if (r[e] >= 0) {
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
    if (threadIdx.x == 0) l_rootbase = l_nroot ? atomicAdd(&counters[CTR_NROOT], l_nroot) : 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) if (myslot[e] >= 0) rootlist[l_rootbase + myslot[e]] = ii[e];
}

// interop with the tile kernels (cl_set_traversal levels 1 and 2): the lists' results at their positions of the layout
__global__ void k_scatter_by_pos(const int* __restrict__ cnt_ptr, const int* __restrict__ pos, const int* __restrict__ val, int* __restrict__ out)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt_ptr[0]) out[pos[k]] = val[k];
}

// ------------------------------------------------------------------------------------------
// k_border_w: the border rule on the walker list
//   variant 2 (R2): the lowest-key adjacent component (first come, cDBSCAN2.py:130,212,352)
//   variant 1 (R1): max over adjacent components whose START POINT is a neighbour (unconditional overwrite,
//                   cDBSCAN.py:172-173), else the lowest-key adjacent component (first come, cDBSCAN.py:179-182)
// wowner[w] = root of the owning component (| OWNER_CONTESTED if the walker has more than one adjacent component), -1 = noise
// ------------------------------------------------------------------------------------------
// A workgroup owns a RANGE OF POSITIONS of the layout: its walkers are a contiguous slice of the walker list, the cores
// around them (HC cores beyond the range on both sides: the windows one strip below / above) a contiguous slice of the core
// list -- staged once as (q, sp) pairs + roots.  A walker's windows start at the rank of K2's hints (no strip table, no search);
// "still in the neighbour strip and inside the q window" is a predicate of the staged pair.  Every staged candidate is a core:
// nothing to step over.
template <int NT, int HC>
__global__ void __launch_bounds__(256)
k_border_w(GridParams g, int ntiles, const int* __restrict__ strip_start, const int* __restrict__ lcnt,
           const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank, const int* __restrict__ wgrank,
           const int2* __restrict__ cpair, const int* __restrict__ croot, const int* __restrict__ ckey,
           const int* __restrict__ cstrip, const int2* __restrict__ wpair, const int* __restrict__ wpos,
           const int* __restrict__ wenc, const int* __restrict__ compkey, const int* __restrict__ ncore,
           int* __restrict__ wowner, int* __restrict__ bsize, int* __restrict__ usize, int* __restrict__ clist, int* __restrict__ counters)
{
    constexpr int WIN = NT + 2 * HC;
    __shared__ int2 lw[WIN];
    __shared__ int lx[WIN];
    const int M = strip_start[g.S];
    const int C = lcnt[0];
    const int tile = ltile_of_block(blockIdx.x);
    const int t0 = tile * NT;
    if (tile >= ntiles || t0 >= M) return;
    const int g0 = t0 >> 6, g1 = (t0 + NT) >> 6;         // (NT is a multiple of 64; the rank arrays reach one group behind the last tile of k_classify)
    const int w0 = wgrank[g0], w1 = wgrank[g1];
    if (w0 == w1) return;
    const int clo = max(cgrank[g0] - HC, 0), chi = min(cgrank[g1] + HC, C);      // staged cores [clo, chi): at most NT + 2 HC
    for (int k = threadIdx.x; k < chi - clo; k += 256) { lw[k] = cpair[clo + k]; lx[k] = croot[clo + k]; }
    __syncthreads();
    const bool v1 = g.variant == CL_VARIANT_CDBSCAN1;
    const int lane = threadIdx.x & 63;
    auto pair_at = [&](int j) { return (j >= clo && j < chi) ? lw[j - clo] : cpair[j]; };
    auto root_at = [&](int j) { return (j >= clo && j < chi) ? lx[j - clo] : croot[j]; };
    for (int h0 = w0; h0 < w1; h0 += 256) {
        const int h = h0 + (int)threadIdx.x;
        const bool act = h < w1;
        int o = -1;
        bool contested = false;
        if (act) {
            const int2 me = wpair[h];
            const int pos = wpos[h], enc = wenc[h];
            const int qlo = me.x - g.eps, qhi = me.x + g.eps;
            const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
            const int plo = me.y - g.peps, phi = me.y + g.peps;
            int bestk = INT_MAX, best = -1, tk = -1, tbest = -1, lastr = -1, lastk = 0, first = -1;
            auto see = [&](int j, int r) {
                if (first < 0) first = r; else if (r != first) contested = true;
                int k;
                if (r == lastr) k = lastk; else { k = compkey[r]; lastr = r; lastk = k; }
                if (k < bestk) { bestk = k; best = r; }
                if (v1 && ckey[j] == k && k > tk) { tk = k; tbest = r; }     // j is its component's start point
            };
            const bool hinted = enc < 0 && ((unsigned)enc & K2H_NONE) != K2H_NONE;
            const int c1 = core_rank(cmask, cgrank, pos);                    // the first core behind the walker
            int ca, cb;
            if (hinted) {
                ca = core_rank(cmask, cgrank, pos - (int)((unsigned)enc & K2H_MASK));
                cb = core_rank(cmask, cgrank, pos + (int)(((unsigned)enc >> K2H_BITS) & K2H_MASK));
            } else {
                // no hints (variant-independent: minPts outside 2..128, pile-ups, hints that left their fields): the cores-only strip table
                const int s = me.y >> g.rbits;
                ca = s > 0 ? lower_bound_pairs(cpair, cstrip[s - 1], cstrip[s], qlo) : 0;
                cb = lower_bound_pairs(cpair, cstrip[s + 1], cstrip[min(s + 2, g.S)], qlo);
                if (s == 0) ca = C;                                          // (no strip below: an empty walk)
            }
            // own strip.  Variant 2: all cores on ONE side of the walker inside its q window are within eps of each other (same
            // strip, q inside one eps) -- one component: the nearest core on either side stands for all of them.  Variant 1 needs
            // every neighbour (its start-point rule looks at single PETs).
            if (!v1) {
                if (c1 > 0) { const int2 p = pair_at(c1 - 1); if (p.y >= pbeg && p.x >= qlo) see(c1 - 1, root_at(c1 - 1)); }
                if (c1 < C) { const int2 p = pair_at(c1); if (p.y < pend && p.x <= qhi) see(c1, root_at(c1)); }
            } else {
                for (int j = c1 - 1; j >= 0; --j) { const int2 p = pair_at(j); if (!(p.y >= pbeg && p.x >= qlo)) break; see(j, root_at(j)); }
                for (int j = c1; j < C; ++j) { const int2 p = pair_at(j); if (!(p.y < pend && p.x <= qhi)) break; see(j, root_at(j)); }
            }
            // one strip below: sp can only be too low; one strip above: only too high.  Four candidates per round.
            for (int j = ca; j < C; j += 4) {
                int2 p[4]; int r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int idx = min(j + k, C - 1); p[k] = pair_at(idx); r[k] = root_at(idx); }
                bool out = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (out || j + k >= C || !((p[k].y < pbeg) & (p[k].x <= qhi))) { out = true; continue; }
                    if (p[k].y >= plo) see(j + k, r[k]);
                }
                if (out) break;
            }
            for (int j = cb; j < C; j += 4) {
                int2 p[4]; int r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int idx = min(j + k, C - 1); p[k] = pair_at(idx); r[k] = root_at(idx); }
                bool out = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (out || j + k >= C || !((p[k].y < pend2) & (p[k].x <= qhi))) { out = true; continue; }
                    if (p[k].y <= phi) see(j + k, r[k]);
                }
                if (out) break;
            }
            o = (v1 && tbest >= 0) ? tbest : best;
            wowner[h] = o < 0 ? -1 : (contested ? (o | OWNER_CONTESTED) : o);
        }
        // counts per owning component, reduced over the lanes of the wave that share the owner.  Only components that are not
        // already >= minPts on their cores need them (release rule of variant 2, drop rule of variant 1).
        const bool cnt_me = o >= 0 && ncore[o >= 0 ? o : 0] < g.minPts;
        {
            // only a component that is not live on its cores alone can end up uncertain: its CONTESTED walkers are all
            // k_emit_records_w has to look at -- they are listed here (one atomic per wave)
            const bool want = cnt_me && contested;
            const unsigned long long wb = __ballot(want);
            if (wb) {
                const int firstl = __ffsll((long long)wb) - 1;
                int lbase = 0;
                if (lane == firstl) lbase = atomicAdd(&counters[CTR_NFLAG], __popcll(wb));
                lbase = __builtin_amdgcn_readlane(lbase, firstl);
                if (want) clist[lbase + lane_rank(wb)] = h;
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

// ---- variant 2 release rule (cDBSCAN2.py:180-183): the records of the contested walkers --------------------------
// The contested walkers that k_border_w listed (walkers of components that are not live on their cores alone), one WAVE each.
// Only one whose first-come owner is UNCERTAIN can change hands (k_resolve_release walks a record's components in key order
// and a live one ends the walk): most listed walkers leave after that test.  The others collect their adjacent components:
// four walks over the core array (own strip downwards / upwards, one strip below, one above), 64 cores per round, every
// DISTINCT root looked at once.
__global__ void __launch_bounds__(TPB)
k_emit_records_w(GridParams g, const int* __restrict__ lcnt, const unsigned long long* __restrict__ cmask,
                 const int* __restrict__ cgrank, const int2* __restrict__ cpair, const int* __restrict__ croot,
                 const int* __restrict__ cstrip, const int2* __restrict__ wpair, const int* __restrict__ wpos,
                 const int* __restrict__ wenc, const int* __restrict__ compkey, const int* __restrict__ state,
                 const int* __restrict__ wowner, Rec* __restrict__ recs, int rec_cap, int* __restrict__ counters,
                 const int* __restrict__ clist)
{
    if (counters[CTR_NU] == 0) return;
    const int nlist = counters[CTR_NFLAG];
    const int C = lcnt[0];
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * (TPB / 64);
    const int wave = blockIdx.x * (TPB / 64) + (int)(threadIdx.x >> 6);
    for (int k0 = 0; k0 < nlist; k0 += 64 * nwaves) {
        const int kl = k0 + lane * nwaves + wave;
        const int ci = kl < nlist ? clist[kl] : -1;
        const int co = ci >= 0 ? wowner[ci] : -1;
        unsigned long long todo = __ballot(ci >= 0 && state[owner_root(co) >= 0 ? owner_root(co) : 0] == ST_UNKNOWN && co >= 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int h = __builtin_amdgcn_readlane(ci, src);                // (wave-uniform from here on)
            const int2 me = wpair[h];
            const int pos = wpos[h], enc = wenc[h];
            const int qlo = me.x - g.eps, qhi = me.x + g.eps;
            const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
            const int plo = me.y - g.peps, phi = me.y + g.peps;
            const int c1 = core_rank(cmask, cgrank, pos);
            int ca, cb;
            if (enc < 0 && ((unsigned)enc & K2H_NONE) != K2H_NONE) {
                ca = core_rank(cmask, cgrank, pos - (int)((unsigned)enc & K2H_MASK));
                cb = core_rank(cmask, cgrank, pos + (int)(((unsigned)enc >> K2H_BITS) & K2H_MASK));
            } else {
                const int s = me.y >> g.rbits;
                ca = s > 0 ? lower_bound_pairs(cpair, cstrip[s - 1], cstrip[s], qlo) : C;
                cb = lower_bound_pairs(cpair, cstrip[s + 1], cstrip[min(s + 2, g.S)], qlo);
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
            // walk w: 0 = own strip downwards from c1 - 1, 1 = own strip upwards from c1, 2 = one strip below from ca, 3 = one above from cb
            const int start[4] = {c1 - 1, c1, ca, cb};
            auto more = [&](int w, int q, int p) {
                return w == 0 ? ((p >= pbeg) & (q >= qlo)) : w == 1 ? ((p < pend) & (q <= qhi)) : w == 2 ? ((p < pbeg) & (q <= qhi)) : ((p < pend2) & (q <= qhi));
            };
            auto acc = [&](int w, int p) { return w == 2 ? p >= plo : (w == 3 ? p <= phi : true); };
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                bool on = true;
                for (int rnd = 0; on; ++rnd) {
                    const int j = w == 0 ? start[w] - lane - 64 * rnd : start[w] + lane + 64 * rnd;
                    const bool in = j >= 0 && j < C;
                    const int2 p = in ? cpair[j] : make_int2(0, 0);
                    const int r = in ? croot[j] : -1;
                    const bool ok = in && more(w, p.x, p.y);
                    const int rv = (ok && acc(w, p.y)) ? r : -1;
                    unsigned long long pending = __ballot(rv >= 0);
                    while (pending) {
                        const int R = __builtin_amdgcn_readlane(rv, __ffsll((long long)pending) - 1);
                        pending &= ~__ballot(rv == R);
                        see(R);
                    }
                    on = __ballot(ok) == ~0ull;
                }
            }
            if (lane == 0) {
                if (overflow) atomicExch(&counters[CTR_OVERFLOW], 1);
                if (any_u) {
                    const int idx = atomicAdd(&counters[CTR_NREC], 1);
                    if (idx >= rec_cap) atomicExch(&counters[CTR_OVERFLOW], 2);
                    else {
                        Rec rec; rec.pt = h;
                        for (int q = 0; q < 4; ++q) rec.r[q] = rr[q];
                        recs[idx] = rec;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_final_lists: labels, cluster table and the distance list of the run, from the two lists
// ------------------------------------------------------------------------------------------
// Item k < C is core k, item C + w is walker w.  llab[k] / ldist[k] = label and in-strip coordinate q of item k (what the
// distance statistics K7 read: d = q + V0; noise never entered a list); labels[row] = label only for labelled items (the
// caller has filled the array with -1: .labels of the reference holds clustered points only, cDBSCAN2.py:186-191); the
// cluster table {minX, maxX, minY, maxY, count} by the two-level reduce-by-key of cl_table.h (pipe.py:78-102).
#define LF_CHUNKS 4
__global__ void __launch_bounds__(BIGTPB)
k_final_lists(GridParams g, const int* __restrict__ lcnt, const int2* __restrict__ cpair, const int* __restrict__ cpos,
              const int* __restrict__ croot, const int2* __restrict__ wpair, const int* __restrict__ wpos,
              const int* __restrict__ wowner, const int* __restrict__ rlabel, const u32* __restrict__ srow,
              int* __restrict__ labels, int* __restrict__ llab, int* __restrict__ ldist, Table t)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int C = lcnt[0], L = C + lcnt[1];
    int idx[LF_CHUNKS], own[LF_CHUNKS], lab[LF_CHUNKS], x[LF_CHUNKS], y[LF_CHUNKS], pos[LF_CHUNKS];
    int2 pr[LF_CHUNKS];
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) {
        idx[ch] = (blockIdx.x * LF_CHUNKS + ch) * BIGTPB + threadIdx.x;
        const int k = idx[ch];
        own[ch] = -1; pos[ch] = 0; pr[ch] = make_int2(0, 0);
        if (k < C) { own[ch] = croot[k]; pr[ch] = cpair[k]; pos[ch] = cpos[k]; }
        else if (k < L) { own[ch] = owner_root(wowner[k - C]); pr[ch] = wpair[k - C]; pos[ch] = wpos[k - C]; }
    }
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) lab[ch] = own[ch] >= 0 ? rlabel[own[ch]] : -1;
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) {
        const int k = idx[ch];
        x[ch] = 0; y[ch] = 0;
        if (k < L) {
            llab[k] = lab[ch];
            ldist[k] = pr[ch].x;
            if (lab[ch] >= 0) {
                if (labels) labels[srow[pos[ch]]] = lab[ch];
                // X = (v - a) / 2, Y = (v + a) / 2 exactly (v and a have equal parity)
                const int spv = pr[ch].y;
                const int pp = ((spv >> g.rbits) + g.s0) * g.eps + (spv & (g.peps - 1)) + g.A0, qq = pr[ch].x + g.V0;
                const int a = g.swap ? qq : pp, v = g.swap ? pp : qq;
                x[ch] = (v - a) / 2; y[ch] = (v + a) / 2;
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) table_accumulate(t, h, lab[ch], x[ch], y[ch]);
    table_flush(t, h);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ListBufs {
    unsigned long long *cmask, *wmask, *hmask;
    int *cgloc, *wgloc, *cgrank, *wgrank;
    int *bsum, *boff, *lcnt;
};
static int list_bufs(cl_chrom* c, const GridParams& g, int nm, ListBufs* b)
{
    const size_t n = (size_t)c->n;
    const size_t ngrp_cap = n / 64 + LG + 4;             // groups of any run of the handle (+ the one behind the last tile)
    const size_t nblk_cap = n / LT + 4;
    int rc;
    if ((rc = c->l_mask.ensure(3 * ngrp_cap * 8)) || (rc = c->l_rank.ensure(4 * ngrp_cap * 4)) ||
        (rc = c->l_blk.ensure((4 * (nblk_cap + 1) + 8) * 4)) || (rc = c->l_cstrip.ensure(((size_t)g.S + 4) * 4)) ||
        (rc = c->l_wpos.ensure(n * 4)) || (rc = c->l_wenc.ensure(n * 4)) || (rc = c->l_dist.ensure(n * 4))) return rc;
    b->cmask = c->l_mask.as<unsigned long long>(); b->wmask = b->cmask + ngrp_cap; b->hmask = b->wmask + ngrp_cap;
    b->cgloc = c->l_rank.as<int>(); b->wgloc = b->cgloc + ngrp_cap; b->cgrank = b->wgloc + ngrp_cap; b->wgrank = b->cgrank + ngrp_cap;
    const int nblk = nblocks(nm, LT);
    b->bsum = c->l_blk.as<int>(); b->boff = b->bsum + 2 * (nblk + 1); b->lcnt = c->l_blk.as<int>() + 4 * (nblk_cap + 1);
    return CL_OK;
}

int lists_build(cl_chrom* c, const GridParams& g, int nm, ListRun* out)
{
    ListBufs b;
    int rc = list_bufs(c, g, nm, &b);
    if (rc) return rc;
    const int nblk = nblocks(nm, LT);
    const bool v2 = g.variant == CL_VARIANT_CDBSCAN2;
    int* ticket = c->counters.as<int>() + CTR_TICKET_A;
    ListRun L{};
    L.lcnt = b.lcnt; L.cmask = b.cmask; L.wmask = b.wmask; L.cgrank = b.cgrank; L.wgrank = b.wgrank;
    L.cpair = c->keys_in.as<int2>(); L.wpair = c->keys_out.as<int2>();     // (the sort buffers are dead behind the layout)
    L.cpos = c->head.as<int>(); L.ckey = c->hi.as<int>();
    L.wpos = c->l_wpos.as<int>(); L.wenc = c->l_wenc.as<int>(); L.cstrip = c->l_cstrip.as<int>();
    if (v2) {
        hipLaunchKernelGGL(k_classify<true>, dim3(nblk), dim3(256), 0, c->stream, g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws,
                           (const u32*)c->srow, b.cmask, b.wmask, b.hmask, b.cgloc, b.wgloc, b.bsum, b.boff, ticket, b.lcnt, c->cellfirst.as<int>());
        hipLaunchKernelGGL(k_make_lists<true>, dim3(nblk), dim3(256), 0, c->stream, g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws,
                           (const u32*)c->srow, (const unsigned long long*)b.cmask, (const unsigned long long*)b.wmask, (const unsigned long long*)b.hmask,
                           (const int*)b.cgloc, (const int*)b.wgloc, (const int*)b.boff, (const int*)c->cellfirst.as<int>(), b.cgrank, b.wgrank,
                           L.cpair, L.cpos, L.ckey, L.wpair, L.wpos, L.wenc, L.cstrip);
    } else {
        hipLaunchKernelGGL(k_classify<false>, dim3(nblk), dim3(256), 0, c->stream, g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws,
                           (const u32*)c->srow, b.cmask, b.wmask, b.hmask, b.cgloc, b.wgloc, b.bsum, b.boff, ticket, b.lcnt, c->cellfirst.as<int>());
        hipLaunchKernelGGL(k_make_lists<false>, dim3(nblk), dim3(256), 0, c->stream, g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws,
                           (const u32*)c->srow, (const unsigned long long*)b.cmask, (const unsigned long long*)b.wmask, (const unsigned long long*)b.hmask,
                           (const int*)b.cgloc, (const int*)b.wgloc, (const int*)b.boff, (const int*)c->cellfirst.as<int>(), b.cgrank, b.wgrank,
                           L.cpair, L.cpos, L.ckey, L.wpair, L.wpos, L.wenc, L.cstrip);
    }
    // chains (the core count is only known on the device: the grids are sized by the PETs of the run)
    hipLaunchKernelGGL(k_chain_c, dim3(nblocks(nm, 256 * CH_PER)), dim3(256), 0, c->stream, g, L.lcnt, (const int2*)L.cpair, c->chainflag.as<int>(),
                       c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                       c->lo.as<int>());
    *out = L;
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

// croot: c->root (levels >= 2) or, while the tile kernels still read roots by position (level 1), the distance-list buffer
static int* croot_of(cl_chrom* c) { return c->traversal >= 2 ? c->root.as<int>() : c->l_dist.as<int>(); }

int lists_union_flatten(cl_chrom* c, const GridParams& g, int nm, const ListRun& L)
{
    constexpr int UNT = 1024;
    const int nt = nblocks(nm, UNT);
    // the union walk looks one strip back, i.e. about one strip's cores in front of the core: the halo follows the mean strip population
    if ((long long)c->n > 80LL * g.S)
        hipLaunchKernelGGL((k_union_c<UNT, 512>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, g, nt, L.lcnt, (const int2*)L.cpair, (const int*)c->chainflag.as<int>(),
                           (const int*)L.cstrip, (const int*)c->lo.as<int>(), c->parent.as<int>());
    else
        hipLaunchKernelGGL((k_union_c<UNT, 128>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, g, nt, L.lcnt, (const int2*)L.cpair, (const int*)c->chainflag.as<int>(),
                           (const int*)L.cstrip, (const int*)c->lo.as<int>(), c->parent.as<int>());
    hipLaunchKernelGGL(k_flatten_c, dim3(nblocks(nm, BIGTPB * FLAT_PER)), dim3(BIGTPB), 0, c->stream, L.lcnt, (const int*)c->chainflag.as<int>(),
                       (const int*)c->parent.as<int>(), (const int*)L.ckey, croot_of(c), c->compkey.as<int>(), c->ncore.as<int>(), c->rootlist.as<int>(),
                       c->counters.as<int>());
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

int lists_scatter_root(cl_chrom* c, int nm, const ListRun& L)
{
    HIP_TRY(hipMemsetAsync(c->root.p, 0xFF, (size_t)nm * 4, c->stream));
    LAUNCH(k_scatter_by_pos, nm, L.lcnt, (const int*)L.cpos, (const int*)croot_of(c), c->root.as<int>());
    return CL_OK;
}

int lists_border(cl_chrom* c, const GridParams& g, int nm, const ListRun& L)
{
    constexpr int BNT = 2048, BHC = 256;
    const int nt = nblocks(nm, BNT);
    hipLaunchKernelGGL((k_border_w<BNT, BHC>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, g, nt, (const int*)c->w_strip, L.lcnt, L.cmask, L.cgrank, L.wgrank,
                       (const int2*)L.cpair, (const int*)croot_of(c), (const int*)L.ckey, (const int*)L.cstrip, (const int2*)L.wpair, (const int*)L.wpos,
                       (const int*)L.wenc, (const int*)c->compkey.as<int>(), (const int*)c->ncore.as<int>(), c->owner.as<int>(), c->bsize.as<int>(),
                       c->usize.as<int>(), c->chainflag.as<int>() /* clist: the chain ids are dead */, c->counters.as<int>());
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

int lists_emit_records(cl_chrom* c, const GridParams& g, int nm, const ListRun& L)
{
    (void)nm;
    hipLaunchKernelGGL(k_emit_records_w, dim3(2048), dim3(TPB), 0, c->stream, g, L.lcnt, L.cmask, L.cgrank, (const int2*)L.cpair, (const int*)croot_of(c),
                       (const int*)L.cstrip, (const int2*)L.wpair, (const int*)L.wpos, (const int*)L.wenc, (const int*)c->compkey.as<int>(),
                       (const int*)c->state.as<int>(), (const int*)c->owner.as<int>(), c->recs.as<Rec>(), (int)c->n, c->counters.as<int>(),
                       (const int*)c->chainflag.as<int>());
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

// level 2: the tile kernel k_final_labels reads owners by position
int lists_scatter_owner(cl_chrom* c, int nm, const ListRun& L)
{
    int* opos = c->l_dist.as<int>();
    HIP_TRY(hipMemsetAsync(opos, 0xFF, (size_t)nm * 4, c->stream));
    LAUNCH(k_scatter_by_pos, nm, L.lcnt, (const int*)L.cpos, (const int*)croot_of(c), opos);
    LAUNCH(k_scatter_by_pos, nm, L.lcnt + 1, (const int*)L.wpos, (const int*)c->owner.as<int>(), opos);
    return CL_OK;
}

int lists_final(cl_chrom* c, const GridParams& g, int nm, const ListRun& L, bool rows)
{
    cl_chrom::Slot& sl = c->slot[c->cur];
    hipLaunchKernelGGL(k_final_lists, dim3(nblocks(nm, BIGTPB * LF_CHUNKS)), dim3(BIGTPB), 0, c->stream, g, L.lcnt, (const int2*)L.cpair, (const int*)L.cpos,
                       (const int*)croot_of(c), (const int2*)L.wpair, (const int*)L.wpos, (const int*)c->owner.as<int>(), (const int*)c->chainhead.as<int>(),
                       (const u32*)c->srow, rows ? sl.labels.as<int>() : (int*)nullptr, sl.slab.as<int>(), c->l_dist.as<int>(), make_table(c));
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

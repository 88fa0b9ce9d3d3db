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
//   k_classify_q / k_make_lists_q   the same two passes straight over the BASE layout of the eps (traversal level 4: no compacted copy
//                 of the layout per run; a PET is in the run if q >= the cut's threshold, its word comes from the count cache of the
//                 eps or, inside the cut band, from k_band).
//   k_prep_c      the cores-only strip table; k_classify's sums back to zero.
//   k_union_c     cores of one strip within eps in q form a chain = a contiguous run of the core array, and chains are the nodes
//                 of the union-find: tiles of the CORE array with a left halo make their chains in LDS (open flags while staging,
//                 heads and skips by ballots), then the cross-strip core-core edges by lock-free hooking on the chain heads (the
//                 window one strip below holds cores only: nothing to step over).  k_chain_c: the chains as a kernel of their
//                 own (developer A/B).  k_union_overflow: cores whose window is not staged.
//   k_flatten_c   root per core, component key / core count per root, the root list.
//   k_border_q    variant 2's border rule for the walkers; the cores around a tile's position range are a contiguous slice of the
//                 core array, staged once; a walker's windows start at rank(K2 hint); walks capped at two steps, the long ones
//                 queued in LDS and finished by 16 lanes each.  k_border_w: variant 1 (needs every core neighbour).
//   k_emit_records_w, k_final_lists: the release records of variant 2 and labels / cluster table / distance list from the
//                 two lists.
//
// Results are bit-identical to the tile kernels of cloops_hip.hip (cl_set_traversal switches between them; tests compare).
#include "cl_chrom.h"

#ifdef CLOOPS_DEVEL
#define L_ABL(bit) (g.dbg2 & (bit))                     // developer ablation (CLOOPS_DBG2; results invalid)
#else
#define L_ABL(bit) 0
#endif
#ifdef CLOOPS_DEVEL
// developer build: event counts of the list kernels (walkers, walk iterations, ...), summed per wave; cl_debug_lstats reads and clears them
__device__ unsigned long long g_lstat[32];
__device__ __forceinline__ void lstat_add(int slot, int v)
{
    int t = v;
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(&g_lstat[slot], (unsigned long long)t);
}
extern "C" void cl_debug_lstats(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstat), sizeof(g_lstat));
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lstat), z, sizeof(z));
}
#define LSTAT_ON (g.dbg2 & (1 << 30))                  // (the counters are same-address atomics: only when asked for)
#define LSTAT(slot, v) do { if (LSTAT_ON) lstat_add((slot), (v)); } while (0)
#else
#define LSTAT(slot, v) do { } while (0)
#endif
// What a walker keeps of its K2 word: where its q windows in the strips s-1 / s+1 start, as distances in POSITIONS of the
// layout the lists index -- low 16 bits back to the first PET of strip s-1 with q >= q - eps, high 16 bits ahead to the first one
// of strip s+1 (the builders convert K2's hints: cl_common.h "K2W"); LH_NONE = no hints (the cores-only strip table + a search)
#define LH_NONE 0xffffffffu
__device__ __forceinline__ unsigned lh_pack(int w, int shiftA, int shiftB)
{
    if (!(w < 0 && ((unsigned)w & K2H_NONE) != K2H_NONE)) return LH_NONE;
    const int da = (int)((unsigned)w & K2H_MASK) + shiftA, db = (int)(((unsigned)w >> K2H_BITS) & K2H_MASK) + shiftB;
    return ((da >= 0) & (da < 0xffff) & (db >= 0) & (db < 0xffff)) ? ((unsigned)da | ((unsigned)db << 16)) : LH_NONE;
}
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
           int* __restrict__ bsum /* [2][nblk + 1]: cores / walkers of every tile */, int* __restrict__ sup /* [2][nsup]: ... of every 64 tiles (zero on entry) */,
           int* __restrict__ cellfirst)
{
    __shared__ int lmin[V2 ? LT : 1];
    __shared__ unsigned long long l_chead[LG];
    __shared__ int l_cc[LG], l_wc[LG];
    __shared__ int l_hlast;
    const int M = strip_start[g.S];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nmask = ~(g.peps - 1);
    int q[8], sp[8], w[8], pq[8], pp[8];
    u32 row[8];
    if (V2) {
        // every load of the thread in flight at once: its 8 PETs' (q, sp, row); the predecessor of a PET (cell heads look one PET
        // back) sits in the neighbouring lane -- only lane 0 loads it (the padded arrays hold a sentinel in front of index 0)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            const bool in = i < M;
            q[u] = in ? sv[i] : INT_MAX; sp[u] = in ? sa[i] : INT_MAX; row[u] = in ? srow[i] : 0u;
            pq[u] = (in && lane == 0) ? sv[i - 1] : 0; pp[u] = (in && lane == 0) ? sa[i - 1] : 0;
        }
        for (int k = threadIdx.x; k < LT; k += 256) lmin[k] = INT_MAX;
        if (threadIdx.x == 0) l_hlast = -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int a = wave_shr1(q[u]), b = wave_shr1(sp[u]);      // (lane 0 reads its own value back: handled apart)
            if (lane != 0) { pq[u] = a; pp[u] = b; }
        }
    } else {
        const bool need = ws.D != nullptr;               // (a run on the layout its words were made on finds them by position alone)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            q[u] = (need && i < M) ? sv[i] : 0; sp[u] = (need && i < M) ? sa[i] : 0;
        }
    }
    if (L_ABL(1 << 16)) {
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = q[u] & 63;    // (ablation: no word loads)
    } else
    {
        // the K2 words, in two stages with all loads of a stage in flight: the per-strip offset of a run that reads the words of
        // an earlier run of its eps (WordSrc: position i of strip s is position i + D[s] there; the cut band has fresh words), then
        // the words themselves
        const int* src[8]; int idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            src[u] = ws.rc; idx[u] = i < M ? i : 0;
            if (ws.D && i < M) {
                if (q[u] < ws.bandq) src[u] = ws.band;
                else idx[u] = i + ws.D[sp[u] >> ws.rbits];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = src[u][idx[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = t0 + u * 256 + (int)threadIdx.x; w[u] = i < M ? w[u] : 0; }
    }
    if (L_ABL(1 << 20)) {                                // (ablation: the loads alone)
        int acc = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += w[u] + q[u] + sp[u] + (V2 ? (int)row[u] + pq[u] + pp[u] : 0);
        if (acc == 0x7f123456) cgloc[0] = acc;
        return;
    }
    unsigned headbits = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < M;
        const bool core = in && cw_core(w[u], g.minPts);
        const bool walk = in && !core && cw_count(w[u]) > 1;      // (count <= 1: nothing within eps -- most of the background noise ends here)
        const unsigned long long cb = __ballot(core), wb = __ballot(walk);
        const int k = u * 4 + wv, gidx = (t0 >> 6) + k;
        if (lane == 0) { if (!L_ABL(1 << 19)) { cmask[gidx] = cb; wmask[gidx] = wb; } l_cc[k] = __popcll(cb); l_wc[k] = __popcll(wb); }
        if (V2) {
            // a PET starts a rotated cell (strip, q / eps) iff its predecessor lies in an earlier strip or below the cell's lower q
            // edge (cDBSCAN2.py:67-70; variant 2 runs with A0 = V0 = 0)
            const int q0 = in ? div_eps(g, q[u]) * g.eps : 0;
            const bool start = in && (i == 0 || (pp[u] & nmask) != (sp[u] & nmask) || pq[u] < q0);
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
        if (!L_ABL(1 << 19)) (lane < 32 ? cgloc : wgloc)[(t0 >> 6) + k] = incl - v;
        totC = __shfl(incl, 31); totW = __shfl(incl, 63);
    }
    // the tile's totals: a plain store, and two atomics NOBODY WAITS FOR into the sums of its 64-tile superblock -- k_make_lists
    // adds up the superblocks and the tiles in front of its own (a few hundred loads per workgroup).  (First form: block sums as
    // awaited device-scope atomics, a ticket, the last workgroup scans -- 57 of this kernel's 89 us on chr1: ~5000 workgroups each
    // ended in two dependent round trips to memory.)
    if (threadIdx.x == 0 && !L_ABL(1 << 18)) {
        const int nsup = (nblk + 63) / 64 + 1;
        bsum[blk] = totC; bsum[nblk + 1 + blk] = totW;
        if (totC) atomicAdd(&sup[blk >> 6], totC);
        if (totW) atomicAdd(&sup[nsup + (blk >> 6)], totW);
    }
    if (V2 && !L_ABL(256)) {
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
            // the cell of the tile's last PET may go on behind the tile: wave 0 walks it, 64 PETs per round
            const int2 lp = make_int2(sv[tend - 1], sa[tend - 1]);
            const int p0 = lp.y & nmask, qend = div_eps(g, lp.x) * g.eps + g.eps;
            int m = INT_MAX;
            for (int j0 = tend; j0 < M; j0 += 64) {
                const int j = j0 + (int)threadIdx.x;
                bool in = j < M;
                if (in) {
                    const int2 cc = make_int2(sv[j], sa[j]);
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
#ifdef CLOOPS_DEVEL
    if (V2 && L_ABL(256)) {                              // (ablation: a valid key all the same -- the head's own row)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = t0 + u * 256 + (int)threadIdx.x; if (i < M) cellfirst[i] = (int)row[u]; }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// k_make_lists
// ------------------------------------------------------------------------------------------
template <bool V2>
__global__ void __launch_bounds__(256)
k_make_lists(GridParams g, const int* __restrict__ sv, const int* __restrict__ sa, const int* __restrict__ strip_start, WordSrc ws,
             const u32* __restrict__ srow, const unsigned long long* __restrict__ cmask, const unsigned long long* __restrict__ wmask,
             const unsigned long long* __restrict__ hmask, const int* __restrict__ cgloc, const int* __restrict__ wgloc,
             const int* __restrict__ bsum, const int* __restrict__ sup, const int* __restrict__ cellfirst, int* __restrict__ cgrank, int* __restrict__ wgrank,
             int2* __restrict__ cpair, int* __restrict__ cpos, int* __restrict__ ckey, int2* __restrict__ wpair,
             int* __restrict__ wpos, int* __restrict__ wenc, int* __restrict__ lcnt)
{
    __shared__ int l_red[2][4];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // cores / walkers in front of this tile: the 64-tile superblocks in front of its own + the tiles of its own in front of it
    int cbase, wbase;
    {
        const int nsup = (nblk + 63) / 64 + 1, sb = blk >> 6;
        int pc = 0, pw = 0;
        for (int k = threadIdx.x; k < sb; k += 256) { pc += sup[k]; pw += sup[nsup + k]; }
        if (threadIdx.x < 64) { const int k = sb * 64 + (int)threadIdx.x; if (k < blk) { pc += bsum[k]; pw += bsum[nblk + 1 + k]; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pc += __shfl_down(pc, o); pw += __shfl_down(pw, o); }
        if (lane == 0) { l_red[0][wv] = pc; l_red[1][wv] = pw; }
        __syncthreads();
        cbase = l_red[0][0] + l_red[0][1] + l_red[0][2] + l_red[0][3];
        wbase = l_red[1][0] + l_red[1][1] + l_red[1][2] + l_red[1][3];
    }
    if (blk == nblk - 1 && threadIdx.x == 0) {
        // the last tile knows the totals: {C, W} for every kernel behind this one, and the rank of the group behind the last tile
        const int C = cbase + bsum[blk], W = wbase + bsum[nblk + 1 + blk];
        lcnt[0] = C; lcnt[1] = W;
        cgrank[nblk * LG] = C; wgrank[nblk * LG] = W;
#ifdef CLOOPS_DEVEL
        if (LSTAT_ON) { atomicAdd(&g_lstat[8], (unsigned long long)C); atomicAdd(&g_lstat[9], (unsigned long long)W); atomicAdd(&g_lstat[10], (unsigned long long)strip_start[g.S]); atomicAdd(&g_lstat[11], 1ull); }
#endif
    }
    // stage by stage over the thread's 8 PETs (one per 64-PET group of its wave), all loads of a stage in flight together:
    // the groups' masks and ranks (wave-uniform), the pairs, then what a core / a walker needs beyond them
    unsigned long long cbv[8], wbv[8];
    int cgv[8], wgv[8], q[8], sp[8], aux[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int gidx = (t0 >> 6) + u * 4 + wv;
        cbv[u] = cmask[gidx]; wbv[u] = wmask[gidx];
        cgv[u] = cbase + cgloc[gidx]; wgv[u] = wbase + wgloc[gidx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool any = ((cbv[u] | wbv[u]) >> lane) & 1ull;
        q[u] = any ? sv[i] : 0; sp[u] = any ? sa[i] : 0;
        if (lane == 0) { const int gidx = (t0 >> 6) + u * 4 + wv; cgrank[gidx] = cgv[u]; wgrank[gidx] = wgv[u]; }
    }
    {
        // walkers: their K2 word (count + window hints, shifted into this run's layout: WordSrc); variant 1 cores: their input row
        WordSrc::Where at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            const bool isw = (wbv[u] >> lane) & 1ull;
            at[u].src = nullptr; at[u].idx = 0; at[u].dA = 0; at[u].dB = 0;
            if (isw) at[u] = ws.where(i, q[u], sp[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = t0 + u * 256 + (int)threadIdx.x;
            const bool isc = (cbv[u] >> lane) & 1ull;
            aux[u] = at[u].src ? at[u].src[at[u].idx] : ((!V2 && isc) ? (int)srow[i] : 0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (at[u].src) aux[u] = ws.shifted(aux[u], at[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const int gidx = (t0 >> 6) + u * 4 + wv;
        const unsigned long long cb = cbv[u], wb = wbv[u];
        const bool isc = (cb >> lane) & 1ull, isw = (wb >> lane) & 1ull;
        if (isc) {
            const int dst = cgv[u] + __popcll(cb & low_mask(lane));
            int key = aux[u];                             // variant 1: the component's start point is its smallest-row core (cDBSCAN.py:134-137)
            if (V2 && L_ABL(512)) key = (int)srow[i];
            else if (V2) {
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
            }
            cpair[dst] = make_int2(q[u], sp[u]); cpos[dst] = i; ckey[dst] = key;
        } else if (isw) {
            const int dst = wgv[u] + __popcll(wb & low_mask(lane));
            wpair[dst] = make_int2(q[u], sp[u]); wpos[dst] = i; wenc[dst] = (int)lh_pack(aux[u], 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Level 4: the lists straight from the BASE layout
// ------------------------------------------------------------------------------------------
// The base layout of an eps holds ALL rows, sorted (strip, q); a cut (pipe.py:59-62: d = Y - X >= cut, and q IS d) removes a
// PREFIX of every strip.  k_cut_strips leaves, per strip, tab[s] = {first kept base index, end of the cut band, PETs removed};
// the two kernels below classify and compact the KEPT PETs of the base layout by those indices -- no copy of the layout,
// positions are base positions (a removed PET has neither bit, so rank(any position of a removed prefix) = rank(the strip's
// first kept PET): K2's window hints, converted from positions of the layout the words were made on to base positions by adding
// what THAT layout's cut removed from the strip in between, stay valid as they are).
//
// Variant 2's component keys are minima over rotated CELLS (cDBSCAN2.py:117-140: the smallest input row of any PET of the cell).
// A cell is a run of the base layout and the cut removes whole cells plus a part of ONE cell per strip: bkey[b] = the minimum
// of b's cell over ALL rows is a property of the base layout (k_base_keys, once per eps), and k_cut_strips recomputes the one
// cell per strip its cut goes through (fix[s] = {end of that cell, its minimum over the kept PETs}).

// bkey[b] for every base position (variant 2): a tile owns the cells that BEGIN in it -- their minima by LDS atomics, the part of
// its last cell that runs on behind the tile walked (and written) by wave 0.
__global__ void __launch_bounds__(256)
k_base_keys(GridParams g, int npos, const int* __restrict__ bq, const int* __restrict__ bsp, const u32* __restrict__ brow, int* __restrict__ bkey)
{
    __shared__ int lmin[LT];
    __shared__ unsigned long long l_chead[LG];
    __shared__ int l_hlast;
    const int blk = (int)blockIdx.x, t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nmask = ~(g.peps - 1);
    int q[8], sp[8], pq[8], pp[8], pos[8];
    u32 row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < npos;
        q[u] = in ? bq[i] : INT_MAX; sp[u] = in ? bsp[i] : INT_MAX; row[u] = in ? brow[i] : 0u;
        pq[u] = (in && lane == 0) ? bq[i - 1] : 0; pp[u] = (in && lane == 0) ? bsp[i - 1] : 0;      // (the padded arrays hold a sentinel in front of index 0)
    }
    for (int k = threadIdx.x; k < LT; k += 256) lmin[k] = INT_MAX;
    if (threadIdx.x == 0) l_hlast = -1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int a = wave_shr1(q[u]), b = wave_shr1(sp[u]);      // (lane 0 reads its own value back: handled apart)
        if (lane != 0) { pq[u] = a; pp[u] = b; }
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < npos;
        const int q0 = in ? div_eps(g, q[u]) * g.eps : 0;
        const bool start = in && (i == 0 || (pp[u] & nmask) != (sp[u] & nmask) || pq[u] < q0);
        const unsigned long long hb = __ballot(start);
        if (lane == 0) l_chead[u * 4 + wv] = hb;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        pos[u] = -1;
        if (i >= npos) continue;
        const int k = u * 4 + wv;
        const unsigned long long upto = l_chead[k] & ((2ull << lane) - 1ull);
        if (upto) pos[u] = (i - lane) + 63 - __clzll((long long)upto);
        else
            for (int k2 = k - 1; k2 >= 0; --k2) {
                const unsigned long long o2 = l_chead[k2];
                if (o2) { pos[u] = t0 + 64 * k2 + 63 - __clzll((long long)o2); break; }
            }
        if (i == min(t0 + LT, npos) - 1) l_hlast = pos[u];
    }
    // a cell's PETs sit in neighbouring lanes: the wave is cut into runs of one cell, a run is reduced toward its last lane (on the
    // DPP network) and that lane alone goes to the cell's slot -- one LDS atomic per cell and wave instead of one per PET on one address
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const WaveRuns w = wave_runs(pos[u]);
        int m = pos[u] >= t0 ? (int)row[u] : INT_MAX;
        seg_step_min<0x111, 0xf>(w.dist >= 1, m);
        seg_step_min<0x112, 0xf>(w.dist >= 2, m);
        seg_step_min<0x114, 0xf>(w.dist >= 4, m);
        seg_step_min<0x118, 0xf>(w.dist >= 8, m);
        seg_step_min<0x142, 0xa>(w.dist > (lane & 15), m);
        seg_step_min<0x143, 0xc>(w.dist > (lane & 31), m);
        if (w.last && pos[u] >= t0) atomicMin(&lmin[pos[u] - t0], m);
    }
    __syncthreads();
    const int tend = t0 + LT;
    if (threadIdx.x < 64 && tend < npos && l_hlast >= t0) {
        // the cell of the tile's last PET may go on behind the tile: its minimum over that part, then the finished minimum written
        // to that part (the tiles behind leave the PETs of a cell that began in front of them alone)
        const int2 lp = make_int2(bq[tend - 1], bsp[tend - 1]);
        const int p0 = lp.y & nmask, qend = div_eps(g, lp.x) * g.eps + g.eps;
        int m = INT_MAX, jend = tend;
        for (int j0 = tend; j0 < npos; j0 += 64) {
            const int j = j0 + (int)threadIdx.x;
            bool in = j < npos;
            if (in) in = (bsp[j] & nmask) == p0 && bq[j] < qend;
            if (in) m = min(m, (int)brow[j]);
            const unsigned long long bal = __ballot(in);
            jend = j0 + __popcll(bal);                   // (the cell's PETs are contiguous: the set lanes are a prefix)
            if (bal != ~0ull) break;
        }
        m = dpp_reduce_wave(m, OpMin());
        const int fin = min(m, lmin[l_hlast - t0]);
        if (threadIdx.x == 0) lmin[l_hlast - t0] = fin;
        for (int j = tend + (int)threadIdx.x; j < jend; j += 64) bkey[j] = fin;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        if (i < npos && pos[u] >= t0) bkey[i] = lmin[pos[u] - t0];
    }
}

// The count cache of a level-4 handle lives in BASE-POSITION space: words[b] = the K2 word of base position b with its two hint
// fields counted in base positions.  A run on the base layout itself (no cut) gets that from K2 as it is; the first run of an eps
// under a cut queries a compact copy of the layout (k_cut_copy + k_region_core, as before) and k_words_to_base moves its words to
// their base positions, adding to the hints what the cut removed from the strips in between (pre[s] behind the PET's own
// strip start, pre[s + 1] in front of the strip above).  Every later run of the eps then needs NO per-strip table: a PET is kept
// iff q >= its threshold, has a fresh word (k_band) iff q < bandq, and the cached one otherwise.
__global__ void __launch_bounds__(256)
k_words_to_base(GridParams g, int npos, const int* __restrict__ bsp, const int4* __restrict__ tab, const int* __restrict__ poff,
                const int* __restrict__ run_words, int* __restrict__ words)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= npos) return;
    const int st = min(bsp[b] >> g.rbits, g.S);
    const int4 t = tab[st];
    if (b < t.x) return;                                  // removed by the cut
    int w = run_words[b - poff[st]];
    if (w < 0 && ((unsigned)w & K2H_NONE) != K2H_NONE) {
        const int da = (int)((unsigned)w & K2H_MASK) + t.z, db = (int)(((unsigned)w >> K2H_BITS) & K2H_MASK) + tab[min(st + 1, g.S)].z;
        const bool ok = (da < (int)K2H_MASK) & (db < (int)K2H_MASK);
        w = (int)(((unsigned)w & ~K2H_NONE) | (ok ? ((unsigned)da | ((unsigned)db << K2H_BITS)) : K2H_NONE));
    }
    words[b] = w;
}

// classify: one bit per base position "core" / "walker" (removed PETs: neither), their counts per 64-position group exclusive
// inside the tile, the tile's totals (a plain store) and the totals of its 16-tile superblock (ONE atomic nobody waits for:
// cores in the low, walkers in the high half).
template <bool CUT, bool BAND>
__global__ void __launch_bounds__(256)
k_classify_q(GridParams g, int npos, int thr, int bandq, const int* __restrict__ bq, const int* __restrict__ words,
             const int* __restrict__ band, unsigned long long* __restrict__ cmask, unsigned long long* __restrict__ wmask,
             int* __restrict__ cgloc, int* __restrict__ wgloc, int* __restrict__ bsum, unsigned long long* __restrict__ sup)
{
    __shared__ int l_cc[LG], l_wc[LG];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int q[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int b = t0 + u * 256 + (int)threadIdx.x; q[u] = (CUT && b < npos) ? bq[b] : INT_MAX; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int b = t0 + u * 256 + (int)threadIdx.x;
        const bool alive = b < npos && (!CUT || q[u] >= thr);
        const int* src = (BAND && q[u] < bandq) ? band : words;
        w[u] = alive ? src[b] : 0;                        // (word 0: neither core nor walker)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const bool core = cw_core(w[u], g.minPts) && w[u] != 0;
        const bool walk = !core && cw_count(w[u]) > 1;      // (count <= 1: nothing within eps -- most of the background noise ends here)
        const unsigned long long cb = __ballot(core), wb = __ballot(walk);
        const int k = u * 4 + wv, gidx = (t0 >> 6) + k;
        if (lane == 0) { cmask[gidx] = cb; wmask[gidx] = wb; l_cc[k] = __popcll(cb); l_wc[k] = __popcll(wb); }
    }
    if (blk == nblk - 1 && threadIdx.x == 0) {
        const int ge = nblk * LG;                        // one group behind the last tile: position npos may be its first
        cmask[ge] = 0ull; wmask[ge] = 0ull; cgloc[ge] = 0; wgloc[ge] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int k = lane & 31;
        const int v = lane < 32 ? l_cc[k] : l_wc[k];
        int incl = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up(incl, d, 32); incl += k >= d ? t : 0; }
        (lane < 32 ? cgloc : wgloc)[(t0 >> 6) + k] = incl - v;
        const int totC = __shfl(incl, 31), totW = __shfl(incl, 63);
        if (threadIdx.x == 0 && !L_ABL(1 << 18)) {
            bsum[blk] = totC; bsum[nblk + 1 + blk] = totW;
            const unsigned long long both = (unsigned long long)(unsigned)totC | ((unsigned long long)(unsigned)totW << 32);
            if (both) atomicAdd(&sup[blk >> 4], both);
        }
    }
}

template <bool V2, bool CUT, bool BAND>
__global__ void __launch_bounds__(256)
k_make_lists_q(GridParams g, int npos, int bandq, const int* __restrict__ bq, const int* __restrict__ bsp, const u32* __restrict__ brow,
               const int* __restrict__ bkey, const int2* __restrict__ fix, const int* __restrict__ words, const int* __restrict__ band,
               const unsigned long long* __restrict__ cmask, const unsigned long long* __restrict__ wmask, const int* __restrict__ cgloc,
               const int* __restrict__ wgloc, const int* __restrict__ bsum, const unsigned long long* __restrict__ sup,
               int* __restrict__ cgrank, int* __restrict__ wgrank, int2* __restrict__ cpair, int* __restrict__ cpos,
               int* __restrict__ ckey, int2* __restrict__ wpair, int* __restrict__ wpos, int* __restrict__ wenc, int* __restrict__ lcnt,
               int* __restrict__ parent /* or null: every core its own union-find node (the fused k_union_c needs that before it starts) */)
{
    __shared__ int l_red[2][4];
    const int nblk = (int)gridDim.x, blk = (int)blockIdx.x;
    const int t0 = blk * LT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;      // (a wave-uniform wv -- group words by scalar loads, 63 instead of 109 registers, 7 waves per SIMD instead of 4 -- measured 83 against 80 us; q and sp read for LISTED positions only, behind the masks, with a band mask from k_classify_q: 82 + 6 us more in k_classify_q -- the kernel is bound neither by what is in flight nor by those bytes)
    // A tile is a chain of dependent round trips (masks -> keys / hints; tile sums -> places -> stores), and a chromosome is eight
    // rounds of resident workgroups: the loads are issued as early as their addresses are known -- q and sp of every position
    // with the masks (nearly every 64-byte line holds a listed PET anyway), keys / hints and the rows of the boundary cells as
    // soon as the masks are there, and only then the sums in front of the tile, whose barrier nothing else waits for.
    unsigned long long cbv[8], wbv[8];
    int cgv[8], wgv[8], q[8], sp[8], aux[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int gidx = (t0 >> 6) + u * 4 + wv;
        cbv[u] = cmask[gidx]; wbv[u] = wmask[gidx];
        cgv[u] = cgloc[gidx]; wgv[u] = wgloc[gidx];
        const int b = t0 + u * 256 + (int)threadIdx.x;
        q[u] = b < npos ? bq[b] : 0; sp[u] = b < npos ? bsp[b] : 0;
    }
    {
        // a core's key: variant 1 its input row (the component's start point is its smallest-row core, cDBSCAN.py:134-137);
        // variant 2 the minimum of its rotated cell (bkey; the cell the cut goes through: fix).  A walker's window hints: its word.
        int2 fx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = t0 + u * 256 + (int)threadIdx.x;
            const bool isc = (cbv[u] >> lane) & 1ull, isw = (wbv[u] >> lane) & 1ull;
            const int* wsrc = (BAND && q[u] < bandq) ? band : words;
            aux[u] = L_ABL(1 << 25) ? b : (isw ? wsrc[b] : (isc ? (V2 ? bkey[b] : (int)brow[b]) : 0));
            fx[u] = make_int2(INT_MIN, 0);
            if (V2 && CUT && !L_ABL(1 << 26)) {
                // (the rows of the wave's first and last strip are wave-uniform loads; a strip holds tens to hundreds of PETs)
                const unsigned long long anyb = cbv[u] | wbv[u];
                const int st = isc ? (sp[u] >> g.rbits) : 0;
                if (anyb) {
                    const int s_lo = __builtin_amdgcn_readlane(sp[u], __ffsll((long long)anyb) - 1) >> g.rbits;
                    const int s_hi = __builtin_amdgcn_readlane(sp[u], 63 - __clzll((long long)anyb)) >> g.rbits;
                    const int2 fa = fix[s_lo], fz = fix[s_hi];
                    if (isc) fx[u] = st == s_lo ? fa : (st == s_hi ? fz : fix[st]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = t0 + u * 256 + (int)threadIdx.x;
            const bool isw = (wbv[u] >> lane) & 1ull;
            if (V2 && CUT && b < fx[u].x) aux[u] = fx[u].y;
            if (isw) aux[u] = (int)lh_pack(aux[u], 0, 0);
        }
    }
    int cbase, wbase;
    {
        // cores / walkers in front of this tile: the 16-tile superblocks in front of its own + the tiles of its own in front of it
        const int sb = blk >> 4;
        int pc = 0, pw_ = 0;
        for (int k = threadIdx.x; k < sb; k += 256) { const unsigned long long v = sup[k]; pc += (int)(unsigned)v; pw_ += (int)(unsigned)(v >> 32); }
        if (threadIdx.x < 16) { const int k = sb * 16 + (int)threadIdx.x; if (k < blk) { pc += bsum[k]; pw_ += bsum[nblk + 1 + k]; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pc += __shfl_down(pc, o); pw_ += __shfl_down(pw_, o); }
        if (lane == 0) { l_red[0][wv] = pc; l_red[1][wv] = pw_; }
        __syncthreads();
        cbase = l_red[0][0] + l_red[0][1] + l_red[0][2] + l_red[0][3];
        wbase = l_red[1][0] + l_red[1][1] + l_red[1][2] + l_red[1][3];
    }
    if (blk == nblk - 1 && threadIdx.x == 0) {
        const int C = cbase + bsum[blk], W = wbase + bsum[nblk + 1 + blk];
        lcnt[0] = C; lcnt[1] = W;
        cgrank[nblk * LG] = C; wgrank[nblk * LG] = W;
#ifdef CLOOPS_DEVEL
        if (LSTAT_ON) { atomicAdd(&g_lstat[8], (unsigned long long)C); atomicAdd(&g_lstat[9], (unsigned long long)W); atomicAdd(&g_lstat[11], 1ull); }
#endif
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        cgv[u] += cbase; wgv[u] += wbase;
        if (lane == 0) { const int gidx = (t0 >> 6) + u * 4 + wv; cgrank[gidx] = cgv[u]; wgrank[gidx] = wgv[u]; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int b = t0 + u * 256 + (int)threadIdx.x;
        const unsigned long long cb = cbv[u], wb = wbv[u];
        const bool isc = (cb >> lane) & 1ull, isw = (wb >> lane) & 1ull;
        if (L_ABL(1 << 27)) { if (q[u] + sp[u] + aux[u] == 0x7f123457) ckey[0] = 1; continue; }      // (ablation: no stores)
        if (isc) {
            const int dst = cgv[u] + __popcll(cb & low_mask(lane));
            cpair[dst] = make_int2(q[u], sp[u]); ckey[dst] = aux[u];
            if (cpos) cpos[dst] = b;                     // (only a run with row-aligned labels needs a core's position)
            if (parent) parent[dst] = dst;
        } else if (isw) {
            const int dst = wgv[u] + __popcll(wb & low_mask(lane));
            wpair[dst] = make_int2(q[u], sp[u]); wpos[dst] = b; wenc[dst] = aux[u];
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_chain_c: chains on the core array
// ------------------------------------------------------------------------------------------
// Inside a strip every pair is within eps in the strip coordinate, so cores whose q gaps are <= eps form a CHAIN -- on the core
// array a contiguous run.  chainid[c] = index of the chain's first core (its head: the union-find node of all its cores);
// cskip[c] = a core behind c from which a walk that has met c goes on (see below).  Every component root is a chain head: the per-root accumulators are reset here.
#define CH_PER 4
__global__ void __launch_bounds__(256)
k_chain_c(GridParams g, const int* __restrict__ lcnt, const int2* __restrict__ cpair, int* __restrict__ chainid,
          int* __restrict__ parent, int* __restrict__ compkey, int* __restrict__ ncore, int* __restrict__ bsize,
          int* __restrict__ usize, int* __restrict__ state, int* __restrict__ cskip,
          const int* __restrict__ strip_start, const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank,
          int* __restrict__ cstrip, int* __restrict__ sup, int nsup2, int* __restrict__ ckey_prune /* or null */)
{
    const int C = lcnt[0];
    const int lane = threadIdx.x & 63;
    const int nmask = ~(g.peps - 1);
    {
        // the cores-only strip table: cstrip[s] = cores in front of strip s (cstrip[S] = cstrip[S + 1] = C); and the superblock
        // sums of k_classify go back to zero for the next run (k_make_lists has read them)
        const int M = strip_start[g.S];
        for (int u = blockIdx.x * 256 + (int)threadIdx.x; u <= g.S + 1; u += gridDim.x * 256) {
            const int p = u <= g.S ? strip_start[u] : M;
            cstrip[u] = p < M ? core_rank(cmask, cgrank, p) : C;
        }
        if (blockIdx.x == 0) for (int k = threadIdx.x; k < nsup2; k += 256) sup[k] = 0;
    }
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
        // cskip[c]: where a walk that has met core c may go on -- behind the chain's last core, or at the wave's end if the chain
        // runs on (any core of the chain behind c will do: the walks meet the chain again and skip again)
        const unsigned long long lb = __ballot(last) & (~0ull << lane);
        const int skip = (c - lane) + (lb ? __ffsll((long long)lb) : 64);
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
            cskip[c] = skip;
            // variant 2, keys by cell (level 4): of a cell's cores only the first carries the cell to k_flatten_c (two cores of one
            // cell are always one component) -- the others cost it neither an atomic nor a compare
            if (ckey_prune && c > 0 && (pv[e].y & nmask) == p0 && div_eps(g, pv[e].x) == div_eps(g, me[e].x)) ckey_prune[c] = INT_MAX;
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

// FUSE: the chains are made HERE, from the staged pairs (no k_chain_c, no chain ids / skips read back): open flags by one compare
// with the predecessor in LDS, the head of a core's chain = the nearest open flag at or in front of it (ballots per 64 staged
// cores, the chunks' masks in LDS; a chain that began in front of the staged range: one look-back by the first wave, 64 cores per
// round), where a walk goes on behind a chain = the next open flag.  The tile's own cores get their chain id, skip and per-head
// resets written; parent[c] = c was set for every core by k_prep_c (another tile may unite with a head before its own tile runs).
// A core whose window is not staged (a strip population beyond the halo) would need chain ids of other tiles: it is listed and
// k_union_overflow takes it behind this kernel.
struct HeadReset { int* compkey; int* ncore; int* bsize; int* usize; int* state; };
template <int NT, int HALO, bool FUSE>
__global__ void __launch_bounds__(256)
k_union_c(GridParams g, int ntiles, const int* __restrict__ lcnt, const int2* __restrict__ cpair, int* chainid,
          const int* __restrict__ cstrip, int* cskip, int* parent, HeadReset hr, int* __restrict__ ckey_prune /* or null */,
          int* __restrict__ ovlist, int* __restrict__ counters)
{
    constexpr int WIN = NT + HALO;
    constexpr int NCH = WIN / 64;
    __shared__ int2 lw[WIN];
    __shared__ int lx[WIN];
    __shared__ int lend[WIN];                            // where a walk goes on behind the staged core's chain
    __shared__ unsigned long long l_ob[FUSE ? NCH : 1];
    __shared__ int l_found;
    const int C = lcnt[0];
    const int tile = ltile_of_block(blockIdx.x);
    const int t0 = tile * NT;
    if (tile >= ntiles || t0 >= C) return;
    const int base = t0 - HALO;
    const int lane = threadIdx.x & 63;
    const int nmask = ~(g.peps - 1);
    if (FUSE && threadIdx.x == 0) l_found = -1;
    int pf_tb[NT / 256], pf_b[NT / 256];
    {
        // (every load of the thread in flight before the first LDS store: the rolled loop was WIN / 256 dependent round trips per tile)
        constexpr int SU = (WIN + 255) / 256;
        static_assert(WIN % 64 == 0, "whole waves in the last staging round");
        int2 mv[SU], pvl[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int gi = base + u * 256 + (int)threadIdx.x;
            const bool in = gi >= 0 && gi < C && u * 256 + (int)threadIdx.x < WIN;
            mv[u] = in ? cpair[gi] : (gi < 0 ? make_int2(0, INT_MIN) : make_int2(INT_MAX, INT_MAX));
            pvl[u] = (FUSE && lane == 0 && gi > 0 && gi < C) ? cpair[gi - 1] : make_int2(0, 0);      // the first lane of a wave reads its predecessor
        }
        // the rows of the cores-only strip table the thread's OWN cores will need for their windows (strip s-1 = [cstrip[s-1], cstrip[s])):
        // requested as soon as their pairs are here, in flight while the tile makes its chains
        if (HALO % 256 == 0) {
#pragma unroll
            for (int u2 = 0; u2 < NT / 256; ++u2) {
                const int gi = t0 + u2 * 256 + (int)threadIdx.x;
                const int s = mv[HALO / 256 + u2].y >> g.rbits;
                const bool ok = gi < C && s > 0;
                pf_tb[u2] = ok ? cstrip[s - 1] : 0; pf_b[u2] = ok ? cstrip[s] : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int k = u * 256 + (int)threadIdx.x, gi = base + k;
            if (k >= WIN) continue;                      // (whole waves: WIN is a multiple of 64)
            const bool in = gi >= 0 && gi < C;
            const int2 me = mv[u];
            lw[k] = me;
            if (!FUSE) { lx[k] = in ? chainid[gi] : -1; lend[k] = in ? cskip[gi] : 0; }
            else {
                // does the staged core open a chain?  (entries in front of core 0 never do, entries behind the last core always.)  The
                // predecessor's pair comes from the lane in front
                int2 pv = make_int2(wave_shr1(me.x), wave_shr1(me.y));
                if (lane == 0 && gi > 0 && gi < C) pv = pvl[u];
                const bool o = gi <= 0 ? gi == 0 : (gi >= C || (pv.y & nmask) != (me.y & nmask) || pv.x < me.x - g.eps);
                const unsigned long long ob = __ballot(o);
                if (lane == 0) l_ob[k >> 6] = ob;
            }
        }
    }
    __syncthreads();
    if (FUSE) {
        if (threadIdx.x < 64 && base > 0 && !(l_ob[0] & 1ull)) {
            // the chain of the first staged cores opened in front of the staged range: look back, 64 cores per round (core 0 opens a
            // chain: the loop ends)
            int found = -1;
            for (int k0 = base - 64; found < 0; k0 -= 64) {
                const int j = k0 + lane;
                bool o = false;
                if (j >= 0) {
                    const int2 a = cpair[j];
                    const int2 b = j > 0 ? cpair[j - 1] : make_int2(0, INT_MIN);
                    o = j == 0 || (b.y & nmask) != (a.y & nmask) || b.x < a.x - g.eps;
                }
                const unsigned long long bal = __ballot(o);
                if (bal) found = k0 + 63 - __clzll((long long)bal);
            }
            if (lane == 0) l_found = found;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < WIN; k += 256) {
            const int ch = k >> 6, gi = base + k;
            const unsigned long long ob = l_ob[ch];
            const unsigned long long upto = ob & (~0ull >> (63 - lane));
            int head;
            if (upto) head = base + ch * 64 + 63 - __clzll((long long)upto);
            else {
                head = l_found;
                for (int c2 = ch - 1; c2 >= 0; --c2) { const unsigned long long o2 = l_ob[c2]; if (o2) { head = base + c2 * 64 + 63 - __clzll((long long)o2); break; } }
            }
            const unsigned long long above = lane == 63 ? 0ull : (ob & (~0ull << (lane + 1)));
            int skip;
            if (above) skip = base + ch * 64 + __ffsll((long long)above) - 1;
            else {
                skip = base + WIN;                       // (the chain runs on behind the staged range: any core of it behind this one will do)
                for (int c2 = ch + 1; c2 < NCH; ++c2) { const unsigned long long o2 = l_ob[c2]; if (o2) { skip = base + c2 * 64 + __ffsll((long long)o2) - 1; break; } }
            }
            const bool in = gi >= 0 && gi < C;
            lx[k] = in ? head : -1;
            lend[k] = in ? min(skip, C) : 0;
            if (in && k >= HALO) {
                chainid[gi] = head;
                cskip[gi] = min(skip, C);
                if (head == gi) { hr.compkey[gi] = INT_MAX; hr.ncore[gi] = 0; hr.bsize[gi] = 0; hr.usize[gi] = 0; hr.state[gi] = ST_LIVE; }
                // variant 2, keys by cell (level 4): of a cell's cores only the first carries the cell to k_flatten_c (two cores of one
                // cell are always one component) -- the others cost it neither an atomic nor a compare
                if (ckey_prune && gi > 0) {
                    const int2 me = lw[k], pv = lw[k - 1];
                    if ((pv.y & nmask) == (me.y & nmask) && div_eps(g, pv.x) == div_eps(g, me.x)) ckey_prune[gi] = INT_MAX;
                }
            }
        }
        __syncthreads();
    }
    const int wbeg = max(base, 0);
    LdsPairs w; w.a = lw; w.base = base;
    constexpr int PER = NT / 256;
    // (one copy of the body: unrolled over the thread's cores the kernel was 80 KB of code -- more than the instruction cache two CUs share)
#pragma unroll 1
    for (int u = 0; u < PER; ++u) {
        const int i = t0 + u * 256 + (int)threadIdx.x;
        const bool in = i < C;
        int Bs[LU_MAXB];
#pragma unroll
        for (int k = 0; k < LU_MAXB; ++k) Bs[k] = -1;
        int nb = 0;
        int A = -1;
#ifdef CLOOPS_DEVEL
        int st_it = 0, st_touch = 0, st_glb = 0, st_jump = 0;
#endif
        if (in) {
            const int2 me = lw[i - base];
            A = lx[i - base];
            const int s = me.y >> g.rbits;
            int tb, b;                                    // (strip 0 has nothing below: an empty range)
            if (HALO % 256 == 0) { tb = u == 0 ? pf_tb[0] : pf_tb[NT / 256 - 1]; b = u == 0 ? pf_b[0] : pf_b[NT / 256 - 1]; }
            else { tb = s > 0 ? cstrip[s - 1] : 0; b = s > 0 ? cstrip[s] : 0; }
            if (tb < b) {
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
#ifdef CLOOPS_DEVEL
                    ++st_touch;
#endif
                    if (nb < LU_MAXB) {
#pragma unroll
                        for (int k = 0; k < LU_MAXB; ++k) if (k == nb) Bs[k] = B;
                        ++nb;
                    } else {
                        uf_unite(parent, A, B);            // more chains than slots: unite right away
                    }
                };
                if (L_ABL(1 << 17)) tb = b;              // (ablation: neither search nor walk)
                if (tb >= b) { }
                else if (tb >= wbeg && b - tb <= 2047) {
                    // the window is staged.  Once a chain has been touched the walk jumps behind it (cskip, k_chain_c): a window
                    // covered by one chain costs one candidate instead of all of them.
                    const int len = b - tb;
                    int j;
                    if (len <= 31) j = lds_lower_bound8<5>(w, tb, b, qlo);
                    else if (len <= 63) j = lds_lower_bound8<6>(w, tb, b, qlo);
                    else if (len <= 255) j = lds_lower_bound8<8>(w, tb, b, qlo);
                    else j = lds_lower_bound8<11>(w, tb, b, qlo);
                    if (L_ABL(1 << 16)) j = b;           // (ablation: the search without the walk)
                    // four candidates per round; the first that ends the window (q beyond it) or lies within reach decides: the
                    // walk stops, or notes the chain and goes on behind it
                    while (j < b) {
#ifdef CLOOPS_DEVEL
                        ++st_it;
#endif
                        int2 cv[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) cv[k] = lw[min(j + k, b - 1) - base];
                        int f = 4; bool fh = false;
#pragma unroll
                        for (int k = 3; k >= 0; --k) {
                            const bool bey = (j + k >= b) | (cv[k].x > qhi);
                            const bool hit = cv[k].y >= T;
                            if (bey | hit) { f = k; fh = hit & !bey; }
                        }
                        if (f == 4) { j += 4; continue; }
                        if (!fh) break;
                        const int idx = j + f - base;
                        touch(lx[idx]);
#ifdef CLOOPS_DEVEL
                        ++st_jump;
#endif
                        j = lend[idx];                     // (> j + f)
                    }
                } else if (FUSE) {
                    // (chain ids of other tiles are not there yet: k_union_overflow)
                    ovlist[atomicAdd(&counters[CTR_NOVF], 1)] = i;
                } else {
#ifdef CLOOPS_DEVEL
                    ++st_glb;
#endif
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
                                next = cskip[min(k + e, b - 1)];
                            }
                        }
                        k = next;
                    }
                }
            }
        }
        int st_un = 0;
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
            if (rep && !L_ABL(1 << 15)) { uf_unite(parent, A, B); ++st_un; }      // (ablation bit 15: the walks without the union-find steps)
        }
#ifdef CLOOPS_DEVEL
        LSTAT(12, in ? 1 : 0); LSTAT(13, st_it); LSTAT(14, st_touch); LSTAT(15, st_un); LSTAT(16, st_glb); LSTAT(17, st_jump);
        { int mx = st_it; for (int o2 = 32; o2 > 0; o2 >>= 1) mx = max(mx, __shfl_down(mx, o2)); if (lane == 0 && LSTAT_ON) { atomicAdd(&g_lstat[18], (unsigned long long)mx); atomicAdd(&g_lstat[19], 1ull); }
          if (LSTAT_ON && st_it > 2) atomicAdd(&g_lstat[20], 1ull); if (LSTAT_ON && st_it > 4) atomicAdd(&g_lstat[21], 1ull); if (LSTAT_ON && st_it > 8) atomicAdd(&g_lstat[22], 1ull); }
#else
        (void)st_un;
#endif
    }
}

// what has to be in place before the fused k_union_c: every core its own union-find node (a tile may unite with a head of another
// tile before that tile has run), the cores-only strip table, and k_classify's superblock sums back at zero for the next run
__global__ void k_prep_c(GridParams g, const int* __restrict__ lcnt, const int* __restrict__ strip_start,
                         const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank, int* __restrict__ cstrip,
                         int* __restrict__ sup, int nsup2, int* __restrict__ parent /* or null: k_make_lists_q has done it */)
{
    const int C = lcnt[0];
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (parent && u < C) parent[u] = u;
    if (u <= g.S + 1) {
        const int M = strip_start[g.S];
        const int p = u <= g.S ? strip_start[u] : M;
        cstrip[u] = p < M ? core_rank(cmask, cgrank, p) : C;
    }
    if (u < nsup2) sup[u] = 0;
}
// the cores k_union_c<.., true> could not serve from its staged range (their strip below begins in front of it): the same walk
// over global memory, chain ids and skips of every tile in place by now
__global__ void k_union_overflow(GridParams g, const int* __restrict__ counters, const int* __restrict__ ovlist, const int2* __restrict__ cpair,
                                 const int* __restrict__ chainid, const int* __restrict__ cstrip, const int* __restrict__ cskip, int* parent)
{
    const int nov = counters[CTR_NOVF];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nov; t += gridDim.x * blockDim.x) {
        const int i = ovlist[t];
        const int2 me = cpair[i];
        const int A = chainid[i];
        const int s = me.y >> g.rbits;
        if (s <= 0) continue;
        const int tb = cstrip[s - 1], b = cstrip[s];
        const int qlo = me.x - g.eps, qhi = me.x + g.eps, T = me.y - g.peps;
        int k = lower_bound_pairs(cpair, tb, b, qlo);
        int lastB = -1;
        while (k < b) {
            // (eight candidates per round, their loads in flight together: a window beside a cluster the core is not adjacent to is a
            //  long run of misses, and every one of them would be a round trip to L2)
            int2 cv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cv[e] = cpair[min(k + e, b - 1)];
            int f = 8; bool fh = false;
#pragma unroll
            for (int e = 7; e >= 0; --e) {
                const bool bey = (k + e >= b) | (cv[e].x > qhi);
                const bool hit = cv[e].y >= T;
                if (bey | hit) { f = e; fh = hit & !bey; }
            }
            if (f == 8) { k += 8; continue; }
            if (!fh) break;
            const int B = chainid[k + f];
            if (B != lastB) { uf_unite(parent, A, B); lastB = B; }
            k = cskip[k + f];
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_flatten_c: root per core, component keys and core counts, the root list
// ------------------------------------------------------------------------------------------
//   variant 1: key = smallest input row of a core = the component's start point (cDBSCAN.py:134-137)
//   variant 2: key = smallest cellfirst over the cells holding its cores (cDBSCAN2.py:117-140)
// both arrive as ckey[c] (k_make_lists); two-level reduce-by-key as in k_flatten (cloops_hip.hip)
#ifndef FLC_TPB
#define FLC_TPB BIGTPB
#endif
#ifndef FLC_PER
#define FLC_PER 4          // cores per thread of k_flatten_c
#endif
__global__ void __launch_bounds__(FLC_TPB)
k_flatten_c(const int* __restrict__ lcnt, const int* __restrict__ chainid, const int* __restrict__ parent, const int* __restrict__ ckey,
            int* __restrict__ croot, int* __restrict__ compkey,
            int* __restrict__ ncore, int* __restrict__ rootlist, int* __restrict__ counters, int abl)
{
    __shared__ int hkey[AGG_H], hmin[AGG_H], hcnt[AGG_H];
    __shared__ int l_nroot, l_rootbase;
    for (int k = threadIdx.x; k < AGG_H; k += FLC_TPB) { hkey[k] = -1; hmin[k] = INT_MAX; hcnt[k] = 0; }
    if (threadIdx.x == 0) l_nroot = 0;
    __syncthreads();
    const int C = lcnt[0];
    if ((int)blockIdx.x * FLC_TPB * FLC_PER >= C) return;
    int ii[FLC_PER], r[FLC_PER], key[FLC_PER], x[FLC_PER];
    bool in[FLC_PER];
#pragma unroll
    for (int e = 0; e < FLC_PER; ++e) {
        ii[e] = (blockIdx.x * FLC_PER + e) * FLC_TPB + (int)threadIdx.x;
        in[e] = ii[e] < C;
        x[e] = in[e] ? chainid[ii[e]] : -1;
        key[e] = in[e] ? ckey[ii[e]] : INT_MAX;
    }
    {
        // the union kernel has completed (kernel boundary = coherent): plain loads, all walks of the thread step together
        bool todo = false;
#pragma unroll
        for (int e = 0; e < FLC_PER; ++e) todo |= in[e];
        if (abl & (1 << 21)) todo = false;
        while (todo) {
            int p[FLC_PER];
#pragma unroll
            for (int e = 0; e < FLC_PER; ++e) p[e] = in[e] ? parent[x[e]] : -1;
            todo = false;
#pragma unroll
            for (int e = 0; e < FLC_PER; ++e) { todo |= in[e] && p[e] != x[e]; x[e] = in[e] ? p[e] : x[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < FLC_PER; ++e) {
        r[e] = in[e] ? x[e] : -1;
        if (in[e]) croot[ii[e]] = r[e];
    }
    const int lane = threadIdx.x & 63;
    int myslot[FLC_PER];
#pragma unroll
    for (int e = 0; e < FLC_PER; ++e) {
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
    if (!(abl & (1 << 20)))
#pragma unroll
    for (int e = 0; e < FLC_PER; ++e) {
        if (!__any(r[e] >= 0)) continue;
        // cores follow each other in layout order: the cores of a component come in RUNS (chains; neighbouring chains of a
        // cluster; the inside of a large cluster: the whole wave).  A run of equal roots is reduced toward its last lane on the
        // DPP network (wave_runs / seg_step_min of cl_table.h: row shifts, then the two row broadcasts) and that lane alone
        // goes to the table -- a handful of LDS atomics per wave instead of 64 pairs
        const int rv = r[e];
        const WaveRuns w = wave_runs(rv);
        int mk = key[e];
        seg_step_min<0x111, 0xf>(w.dist >= 1, mk);
        seg_step_min<0x112, 0xf>(w.dist >= 2, mk);
        seg_step_min<0x114, 0xf>(w.dist >= 4, mk);
        seg_step_min<0x118, 0xf>(w.dist >= 8, mk);
        seg_step_min<0x142, 0xa>(w.dist > (lane & 15), mk);
        seg_step_min<0x143, 0xc>(w.dist > (lane & 31), mk);
        if (w.last && rv >= 0) {
            const int cntv = w.dist + 1;
            const int sl = agg_slot(hkey, rv);
            if (sl >= 0) { if (mk != INT_MAX) atomicMin(&hmin[sl], mk); atomicAdd(&hcnt[sl], cntv); }
            else { if (mk != INT_MAX) atomicMin(&compkey[rv], mk); atomicAdd(&ncore[rv], cntv); }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < AGG_H; k += FLC_TPB)
        if (hkey[k] >= 0) {
            atomicMin(&compkey[hkey[k]], hmin[k]);
            atomicAdd(&ncore[hkey[k]], hcnt[k]);
        }
    if (threadIdx.x == 0) l_rootbase = l_nroot ? atomicAdd(&counters[CTR_NROOT], l_nroot) : 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < FLC_PER; ++e) if (myslot[e] >= 0) rootlist[l_rootbase + myslot[e]] = ii[e];
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
#define LB_PER 1                 // walkers per thread and pass (a 2048-position tile holds a few hundred walkers: one pass; more per thread only
                                 // multiplies the code -- the kernel's instruction footprint matters: 64 KB of instruction cache per two CUs)
template <int NT, int HC, bool V1>
__global__ void __launch_bounds__(256)
k_border_w(GridParams g, int ntiles, int npos, const int* __restrict__ lcnt,
           const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank, const int* __restrict__ wgrank,
           const int2* __restrict__ cpair, const int* __restrict__ croot, const int* __restrict__ cskip, const int* __restrict__ ckey,
           const int* __restrict__ cstrip, const int2* __restrict__ wpair, const int* __restrict__ wpos,
           const int* __restrict__ wenc, const int* __restrict__ compkey, const int* __restrict__ ncore,
           int* __restrict__ wowner, int* __restrict__ bsize, int* __restrict__ usize, int* __restrict__ clist, int* __restrict__ counters)
{
    constexpr int WIN = NT + 2 * HC;
    __shared__ int2 lw[WIN];
    __shared__ int2 lx[WIN];                             // (root, first core behind the chain)
    const int C = lcnt[0];
    const int tile = ltile_of_block(blockIdx.x);
    const int t0 = tile * NT;
    if (tile >= ntiles || t0 >= npos) return;
    const int g0 = t0 >> 6, g1 = (t0 + NT) >> 6;         // (NT is a multiple of 64; the rank arrays reach one group behind the last tile of k_classify)
    const int w0 = wgrank[g0], w1 = wgrank[g1];
    if (w0 == w1) return;
    const int clo = max(cgrank[g0] - HC, 0), chi = min(cgrank[g1] + HC, C);      // staged cores [clo, chi): at most NT + 2 HC
    for (int k = threadIdx.x; k < chi - clo; k += 256) { lw[k] = cpair[clo + k]; lx[k] = make_int2(croot[clo + k], V1 ? 0 : cskip[clo + k]); }
    __syncthreads();
    if (L_ABL(4096)) { for (int h = w0 + (int)threadIdx.x; h < w1; h += 256) wowner[h] = -1; return; }
    const int lane = threadIdx.x & 63;
    // The kernel is a chain of dependent round trips per walker (record -> rank words -> walks -> keys): a thread takes LB_PER
    // walkers per pass and goes through the stages with the loads of all of them in flight.
    for (int h0 = w0; h0 < w1; h0 += 256 * LB_PER) {
        int2 me[LB_PER]; int pos[LB_PER], enc[LB_PER];
        bool act[LB_PER];
#pragma unroll
        for (int e = 0; e < LB_PER; ++e) {
            const int h = h0 + e * 256 + (int)threadIdx.x;
            act[e] = h < w1;
            me[e] = act[e] ? wpair[h] : make_int2(0, 0); pos[e] = act[e] ? wpos[h] : 0; enc[e] = act[e] ? wenc[h] : 0;
        }
        // where the walks start: the rank of the walker's own position (the first core behind it) and of K2's two hints
        int c1[LB_PER], ca[LB_PER], cb[LB_PER];
        {
            int gr[LB_PER][3]; unsigned long long gm[LB_PER][3]; int pp[LB_PER][3];
#pragma unroll
            for (int e = 0; e < LB_PER; ++e) {
                const bool hinted = (unsigned)enc[e] != LH_NONE;
                pp[e][0] = pos[e];
                pp[e][1] = hinted ? pos[e] - (int)((unsigned)enc[e] & 0xffffu) : pos[e];
                pp[e][2] = hinted ? pos[e] + (int)((unsigned)enc[e] >> 16) : pos[e];
#pragma unroll
                for (int k = 0; k < 3; ++k) { gr[e][k] = cgrank[pp[e][k] >> 6]; gm[e][k] = cmask[pp[e][k] >> 6]; }
            }
#pragma unroll
            for (int e = 0; e < LB_PER; ++e) {
                c1[e] = gr[e][0] + __popcll(gm[e][0] & low_mask(pp[e][0] & 63));
                ca[e] = gr[e][1] + __popcll(gm[e][1] & low_mask(pp[e][1] & 63));
                cb[e] = gr[e][2] + __popcll(gm[e][2] & low_mask(pp[e][2] & 63));
                const bool hinted = (unsigned)enc[e] != LH_NONE;
                if (act[e] && !hinted) {
                    // no hints (minPts outside 2..128, pile-ups, hints that left their fields): the cores-only strip table
                    const int s = me[e].y >> g.rbits, qlo = me[e].x - g.eps;
                    ca[e] = s > 0 ? lower_bound_pairs(cpair, cstrip[s - 1], cstrip[s], qlo) : C;      // (no strip below: an empty walk)
                    cb[e] = lower_bound_pairs(cpair, cstrip[s + 1], cstrip[min(s + 2, g.S)], qlo);
                }
            }
        }
        int own[LB_PER];                                   // the owner found (root), -1 = none
        bool cont[LB_PER];
        int rr[LB_PER][4];                                 // variant 2: the distinct adjacent components
#pragma unroll
        for (int e = 0; e < LB_PER; ++e) {
            own[e] = -1; cont[e] = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[e][k] = -1;
            if (!act[e] || L_ABL(1024)) continue;
            const int qlo = me[e].x - g.eps, qhi = me[e].x + g.eps;
            const int pbeg = me[e].y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
            const int plo = me[e].y - g.peps, phi = me[e].y + g.peps;
            if (V1) {
                // variant 1 needs EVERY core neighbour: its start-point rule looks at single PETs (cDBSCAN.py:172-173)
                int bestk = INT_MAX, best = -1, tk = -1, tbest = -1, lastr = -1, lastk = 0, first = -1;
                bool contested = false;
                auto see = [&](int j, int r) {
                    if (first < 0) first = r; else if (r != first) contested = true;
                    int k;
                    if (r == lastr) k = lastk; else { k = compkey[r]; lastr = r; lastk = k; }
                    if (k < bestk) { bestk = k; best = r; }
                    if (ckey[j] == k && k > tk) { tk = k; tbest = r; }       // j is its component's start point
                };
                auto pair_at = [&](int j) { int2 v; if (j >= clo && j < chi) v = lw[j - clo]; else v = cpair[j]; return v; };
                auto root_at = [&](int j) { int v; if (j >= clo && j < chi) v = lx[j - clo].x; else v = croot[j]; return v; };
                for (int j = c1[e] - 1; j >= 0; --j) { const int2 p = pair_at(j); if (!((p.y >= pbeg) & (p.x >= qlo))) break; see(j, root_at(j)); }
                for (int j = c1[e]; j < C; ++j) { const int2 p = pair_at(j); if (!((p.y < pend) & (p.x <= qhi))) break; see(j, root_at(j)); }
                if (!L_ABL(2048)) {
                    for (int j = ca[e]; j < C; ++j) { const int2 p = pair_at(j); if (!((p.y < pbeg) & (p.x <= qhi))) break; if (p.y >= plo) see(j, root_at(j)); }
                    for (int j = cb[e]; j < C; ++j) { const int2 p = pair_at(j); if (!((p.y < pend2) & (p.x <= qhi))) break; if (p.y <= phi) see(j, root_at(j)); }
                }
                own[e] = tbest >= 0 ? tbest : best;
                cont[e] = contested;
            } else {
                // variant 2: the DISTINCT adjacent components are all that counts (at most 4: cores of different components are
                // > eps apart and all within eps of the walker); their keys are looked up once, behind the walks
                int r0 = -1, r1 = -1, r2 = -1, r3 = -1;
                auto see = [&](int r) {
                    if ((r == r0) | (r == r1) | (r == r2) | (r == r3)) return;
                    if (r0 < 0) r0 = r; else if (r1 < 0) r1 = r; else if (r2 < 0) r2 = r; else if (r3 < 0) r3 = r;
                    else atomicExch(&counters[CTR_OVERFLOW], 1);             // (beyond the geometric bound: the run fails loudly)
                };
                // own strip: all cores on ONE side of the walker inside its q window are within eps of each other (same strip, q
                // inside one eps) -- one component: the nearest core on either side stands for all of them
                {
                    const int jl = max(c1[e] - 1, 0), jr = max(min(c1[e], C - 1), 0);
                    int2 pl, pr; int rl, rrt;
                    if (jl >= clo && jr < chi) { pl = lw[jl - clo]; pr = lw[jr - clo]; rl = lx[jl - clo].x; rrt = lx[jr - clo].x; }      // (LDS and global paths apart: a
                    else { pl = cpair[jl]; pr = cpair[jr]; rl = croot[jl]; rrt = croot[jr]; }                                            //  select of pointers makes FLAT loads)
                    if (c1[e] > 0 && (pl.y >= pbeg) & (pl.x >= qlo)) see(rl);
                    if (c1[e] < C && (pr.y < pend) & (pr.x <= qhi)) see(rrt);
                }
                // one strip below (sp can only be too low) / above (only too high): from the rank of K2's hint on while the pair
                // is still in that strip and inside the q window.  A core within reach settles its whole chain: the walk goes on
                // behind the chain's last core -- a window next to a cluster costs two or three candidates.
#ifdef CLOOPS_DEVEL
                int n_it = 0, n_glb = 0;
#endif
                auto walk = [&](int j, int pcap, bool below) {
                    while (j < C) {
                        const bool in0 = j >= clo && j + 1 < chi;
#ifdef CLOOPS_DEVEL
                        ++n_it; n_glb += in0 ? 0 : 1;
#endif
                        int2 p0, p1, x0, x1;
                        if (in0) { p0 = lw[j - clo]; p1 = lw[j + 1 - clo]; x0 = lx[j - clo]; x1 = lx[j + 1 - clo]; }
                        else {
                            const int j1 = min(j + 1, C - 1);
                            p0 = cpair[j]; p1 = cpair[j1]; x0 = make_int2(croot[j], cskip[j]); x1 = make_int2(croot[j1], cskip[j1]);
                            if (j + 1 >= C) p1 = make_int2(INT_MAX, INT_MAX);
                        }
                        if (!((p0.y < pcap) & (p0.x <= qhi))) break;
                        if (below ? p0.y >= plo : p0.y <= phi) { see(x0.x); j = x0.y; continue; }
                        if (!((p1.y < pcap) & (p1.x <= qhi))) break;
                        if (below ? p1.y >= plo : p1.y <= phi) { see(x1.x); j = x1.y; continue; }
                        j += 2;
                    }
                };
                if (!L_ABL(2048)) { walk(ca[e], pbeg, true); walk(cb[e], pend2, false); }
                rr[e][0] = r0; rr[e][1] = r1; rr[e][2] = r2; rr[e][3] = r3;
                if (L_ABL(1 << 28) && r0 != 0x7f000001) { rr[e][0] = -1; rr[e][1] = -1; rr[e][2] = -1; rr[e][3] = -1; }      // (ablation: the walks, nothing behind them)
#ifdef CLOOPS_DEVEL
                LSTAT(1, n_it); LSTAT(2, n_glb);
                { int mx = n_it; for (int o2 = 32; o2 > 0; o2 >>= 1) mx = max(mx, __shfl_down(mx, o2)); if ((threadIdx.x & 63) == 0 && LSTAT_ON) { atomicAdd(&g_lstat[6], (unsigned long long)mx); atomicAdd(&g_lstat[7], 1ull); } }
#endif
            }
        }
        // keys and core counts of the adjacent components, all loads in flight; variant 2: the lowest key among the distinct
        // adjacent components owns the walker (first come, cDBSCAN2.py:130,212,352)
        int nco[LB_PER];
        if (!V1) {
            int kk[LB_PER][4], nn[LB_PER][4];
#pragma unroll
            for (int e = 0; e < LB_PER; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int r = rr[e][k]; kk[e][k] = r >= 0 ? compkey[r] : INT_MAX; nn[e][k] = r >= 0 ? ncore[r] : 0; }
#pragma unroll
            for (int e = 0; e < LB_PER; ++e) {
                int best = rr[e][0], bestk = kk[e][0], bn = nn[e][0];
#pragma unroll
                for (int k = 1; k < 4; ++k) if (kk[e][k] < bestk) { bestk = kk[e][k]; best = rr[e][k]; bn = nn[e][k]; }
                own[e] = best; cont[e] = rr[e][1] >= 0; nco[e] = bn;
            }
        } else {
#pragma unroll
            for (int e = 0; e < LB_PER; ++e) nco[e] = own[e] >= 0 ? ncore[own[e]] : 0;
        }
#pragma unroll
        for (int e = 0; e < LB_PER; ++e) {
            const int h = h0 + e * 256 + (int)threadIdx.x;
            const int o = own[e];
            const bool contested = cont[e];
            if (act[e]) wowner[h] = o < 0 ? -1 : (contested ? (o | OWNER_CONTESTED) : o);
#ifdef CLOOPS_DEVEL
            LSTAT(0, act[e] ? 1 : 0); LSTAT(5, (act[e] && o >= 0) ? 1 : 0);
#endif
            // counts per owning component, reduced over the lanes of the wave that share the owner.  Only components that are not
            // already >= minPts on their cores need them (release rule of variant 2, drop rule of variant 1).
            const bool cnt_me = act[e] && o >= 0 && nco[e] < g.minPts;
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
}

// ---- k_border_q: variant 2's border rule with the long walks set aside -------------------------------------------
// k_border_w pays, per wave, the LONGEST of its 64 walkers' walks (2.3-3.8 steps per walker on average, 15-25 for the slowest
// lane: a walker whose q window one strip away lies beside a cluster it is not adjacent to passes all of that cluster's
// cores).  Here a walk stops after `kcap` steps; a walker with a walk left over puts itself -- record, where its walks go
// on, the components seen so far -- into a queue in the part of the staging arrays the tile's cores leave free (cores +
// walkers of a tile <= its positions: room for at least 4 of 9 walkers; who finds it full finishes in place), and afterwards
// every queued walker gets BQ_G lanes that test BQ_G candidates per step (a ballot finds the first one that ends the window or
// is within reach): the slow walks become a few steps of a few waves instead of holding up every wave.  Same results as
// k_border_w<.., false>.
__device__ __forceinline__ void see4(int r, int& r0, int& r1, int& r2, int& r3, int* counters)
{
    if ((r == r0) | (r == r1) | (r == r2) | (r == r3)) return;
    if (r0 < 0) r0 = r; else if (r1 < 0) r1 = r; else if (r2 < 0) r2 = r; else if (r3 < 0) r3 = r;
    else atomicExch(&counters[CTR_OVERFLOW], 1);          // (beyond the geometric bound: the run fails loudly)
}

template <int NT, int HC, int BQ_G>
__global__ void __launch_bounds__(256)
k_border_q(GridParams g, int ntiles, int npos, int kcap, const int* __restrict__ lcnt,
           const unsigned long long* __restrict__ cmask, const int* __restrict__ cgrank, const int* __restrict__ wgrank,
           const int2* __restrict__ cpair, const int* __restrict__ croot, const int* __restrict__ cskip,
           const int* __restrict__ cstrip, const int2* __restrict__ wpair, const int* __restrict__ wpos,
           const int* __restrict__ wenc, const int* __restrict__ compkey, const int* __restrict__ ncore,
           int* __restrict__ wowner, int* __restrict__ bsize, int* __restrict__ usize, int* __restrict__ clist, int* __restrict__ counters)
{
    constexpr int WIN = NT + 2 * HC;
    __shared__ int2 lw[WIN];
    __shared__ int2 lx[WIN];                             // (root, first core behind the chain)
    __shared__ int l_nq;
    const int C = lcnt[0];
    const int tile = ltile_of_block(blockIdx.x);
    const int t0 = tile * NT;
    if (tile >= ntiles || t0 >= npos) return;
    const int g0 = t0 >> 6, g1 = (t0 + NT) >> 6;
    const int w0 = wgrank[g0], w1 = wgrank[g1];
    if (w0 == w1) return;
    const int clo = max(cgrank[g0] - HC, 0), chi = min(cgrank[g1] + HC, C);
    if (threadIdx.x == 0) l_nq = 0;
    // the first 256 walkers of the tile (usually all of them): record, then the rank index at its position and at both hints -- two
    // dependent round trips that need nothing of the staged cores, issued in front of the staging loads instead of behind the barrier
    int2 pf_me = make_int2(0, 0);
    int pf_pos = 0, pf_enc = 0, pf_gr0 = 0, pf_gr1 = 0, pf_gr2 = 0;
    unsigned long long pf_gm0 = 0, pf_gm1 = 0, pf_gm2 = 0;
    {
        const int h = w0 + (int)threadIdx.x;
        if (h < w1) { pf_me = wpair[h]; pf_pos = wpos[h]; pf_enc = wenc[h]; }
        const bool hinted = (unsigned)pf_enc != LH_NONE;
        const int pa = hinted ? pf_pos - (int)((unsigned)pf_enc & 0xffffu) : pf_pos, pb = hinted ? pf_pos + (int)((unsigned)pf_enc >> 16) : pf_pos;
        pf_gr0 = cgrank[pf_pos >> 6]; pf_gr1 = cgrank[pa >> 6]; pf_gr2 = cgrank[pb >> 6];
        pf_gm0 = cmask[pf_pos >> 6]; pf_gm1 = cmask[pa >> 6]; pf_gm2 = cmask[pb >> 6];
    }
    {
        // (every staging load of the thread in flight before the first LDS store: the rolled loop was a chain of round trips per tile)
        constexpr int SU = (WIN + 255) / 256;
        int2 pv[SU]; int rv[SU], sv[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int k = u * 256 + (int)threadIdx.x;
            const bool in = k < chi - clo;
            pv[u] = in ? cpair[clo + k] : make_int2(0, 0); rv[u] = in ? croot[clo + k] : 0; sv[u] = in ? cskip[clo + k] : 0;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int k = u * 256 + (int)threadIdx.x;
            if (k < chi - clo) { lw[k] = pv[u]; lx[k] = make_int2(rv[u], sv[u]); }
        }
    }
    __syncthreads();
    const int qbase = (chi - clo + 1) & ~1;              // queue entry e: lw / lx [qbase + 2 e, + 2) = 8 ints
    const int qcap = (WIN - qbase) >> 1;
    const int lane = threadIdx.x & 63;
    // the walker's place in the output, its owner and the counts per owning component (release rule of variant 2): as k_border_w
    auto finish = [&](bool act, int h, int r0, int r1, int r2, int r3) {
        int kk[4], nn[4];
        const int rr[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int r = act ? rr[k] : -1; kk[k] = r >= 0 ? compkey[r] : INT_MAX; nn[k] = r >= 0 ? ncore[r] : 0; }
        int o = act ? rr[0] : -1, bestk = kk[0], nco = nn[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (kk[k] < bestk) { bestk = kk[k]; o = rr[k]; nco = nn[k]; }
        const bool contested = act && r1 >= 0;
        if (act) wowner[h] = o < 0 ? -1 : (contested ? (o | OWNER_CONTESTED) : o);
        const bool cnt_me = act && o >= 0 && nco < g.minPts;
        {
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
    };
    // ---- pass 0: every walker, walks of at most kcap steps ----
#pragma unroll 1
    for (int i0 = 0; i0 < w1 - w0; i0 += 256) {
        const int h = w0 + i0 + (int)threadIdx.x;
        bool act = h < w1;
        int ja = -1, jb = -1, r0 = -1, r1 = -1, r2 = -1, r3 = -1;
        int2 me = make_int2(0, 0);
        {
            int pos = pf_pos, enc = pf_enc;
            int gr0 = pf_gr0, gr1 = pf_gr1, gr2 = pf_gr2;
            unsigned long long gm0 = pf_gm0, gm1 = pf_gm1, gm2 = pf_gm2;
            me = pf_me;
            if (i0 > 0) {                                 // (a tile with more than 256 walkers: the later rounds load behind the barrier)
                pos = 0; enc = 0; me = make_int2(0, 0);
                if (act) { me = wpair[h]; pos = wpos[h]; enc = wenc[h]; }
            }
            const bool hinted = (unsigned)enc != LH_NONE;
            const int pa = hinted ? pos - (int)((unsigned)enc & 0xffffu) : pos, pb = hinted ? pos + (int)((unsigned)enc >> 16) : pos;
            if (i0 > 0) {
                gr0 = cgrank[pos >> 6]; gr1 = cgrank[pa >> 6]; gr2 = cgrank[pb >> 6];
                gm0 = cmask[pos >> 6]; gm1 = cmask[pa >> 6]; gm2 = cmask[pb >> 6];
            }
            const int c1 = gr0 + __popcll(gm0 & low_mask(pos & 63));
            ja = gr1 + __popcll(gm1 & low_mask(pa & 63));
            jb = gr2 + __popcll(gm2 & low_mask(pb & 63));
            if (act && !hinted) {
                // no hints (minPts outside 2..128, pile-ups, hints that left their fields): the cores-only strip table
                const int s = me.y >> g.rbits, qlo = me.x - g.eps;
                ja = s > 0 ? lower_bound_pairs(cpair, cstrip[s - 1], cstrip[s], qlo) : C;
                jb = lower_bound_pairs(cpair, cstrip[s + 1], cstrip[min(s + 2, g.S)], qlo);
            }
            if (act) {
                // own strip: the nearest core on either side stands for all of its side (same strip, q inside one eps: one component)
                const int qlo = me.x - g.eps, qhi = me.x + g.eps;
                const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps;
                const int jl = max(c1 - 1, 0), jr = max(min(c1, C - 1), 0);
                int2 pl, pr; int rl, rrt;
                if (jl >= clo && jr < chi) { pl = lw[jl - clo]; pr = lw[jr - clo]; rl = lx[jl - clo].x; rrt = lx[jr - clo].x; }
                else { pl = cpair[jl]; pr = cpair[jr]; rl = croot[jl]; rrt = croot[jr]; }
                if (c1 > 0 && (pl.y >= pbeg) & (pl.x >= qlo)) see4(rl, r0, r1, r2, r3, counters);
                if (c1 < C && (pr.y < pend) & (pr.x <= qhi)) see4(rrt, r0, r1, r2, r3, counters);
            } else { ja = -1; jb = -1; }
        }
        const int qhi = me.x + g.eps;
        const int pbeg = me.y & ~(g.peps - 1), pend2 = pbeg + 2 * g.peps;
        const int plo = me.y - g.peps, phi = me.y + g.peps;
        // a walk of at most `cap` steps from core j on: -1 = the window is exhausted, else the core to go on from
        auto walk = [&](int j, int pcap, bool below, int cap) -> int {
            if (j < 0) return -1;
            int it = 0;
            while (j < C) {
                if (it >= cap) return j;
                ++it;
                const bool in0 = j >= clo && j + 1 < chi;
                int2 p0, p1, x0, x1;
                if (in0) { p0 = lw[j - clo]; p1 = lw[j + 1 - clo]; x0 = lx[j - clo]; x1 = lx[j + 1 - clo]; }
                else {
                    const int j1 = min(j + 1, C - 1);
                    p0 = cpair[j]; p1 = cpair[j1]; x0 = make_int2(croot[j], cskip[j]); x1 = make_int2(croot[j1], cskip[j1]);
                    if (j + 1 >= C) p1 = make_int2(INT_MAX, INT_MAX);
                }
                if (!((p0.y < pcap) & (p0.x <= qhi))) return -1;
                if (below ? p0.y >= plo : p0.y <= phi) { see4(x0.x, r0, r1, r2, r3, counters); j = x0.y; continue; }
                if (!((p1.y < pcap) & (p1.x <= qhi))) return -1;
                if (below ? p1.y >= plo : p1.y <= phi) { see4(x1.x, r0, r1, r2, r3, counters); j = x1.y; continue; }
                j += 2;
            }
            return -1;
        };
        // kcap steps of both walks as straight-line predicated code (the walks' branches cost a wave more scalar instructions --
        // exec-mask bookkeeping -- than vector ones): a lane whose walk is over, or whose candidates are not staged, idles
        bool ovf = false;
        auto seep = [&](bool on, int r) {
            const bool add = on & !((r == r0) | (r == r1) | (r == r2) | (r == r3));
            const bool s0 = add & (r0 < 0), s1 = add & !s0 & (r1 < 0), s2 = add & !s0 & !s1 & (r2 < 0), s3 = add & !s0 & !s1 & !s2 & (r3 < 0);
            ovf |= add & !s0 & !s1 & !s2 & !s3;
            r0 = s0 ? r : r0; r1 = s1 ? r : r1; r2 = s2 ? r : r2; r3 = s3 ? r : r3;
        };
        auto pstep = [&](int& j, bool& on, int pcap, bool below) {
            const bool go = on & (j >= clo) & (j + 1 < chi);
            const int k0 = go ? j - clo : 0;
            const int2 p0 = lw[k0], p1 = lw[k0 + 1], x0 = lx[k0], x1 = lx[k0 + 1];
            const bool w0 = (p0.y < pcap) & (p0.x <= qhi), h0 = w0 & (below ? p0.y >= plo : p0.y <= phi);
            const bool w1 = (p1.y < pcap) & (p1.x <= qhi), h1 = w1 & (below ? p1.y >= plo : p1.y <= phi);
            const bool hit0 = go & h0, hit1 = go & w0 & !h0 & h1;
            const bool fin = go & (!w0 | (!h0 & !w1));
            seep(hit0 | hit1, hit0 ? x0.x : x1.x);
            j = hit0 ? x0.y : (hit1 ? x1.y : (go ? j + 2 : j));
            on = on & !fin & (j < C);
        };
        bool ona = act & (ja >= 0) & (ja < C), onb = act & (jb >= 0) & (jb < C);
#pragma unroll 1
        for (int it = 0; it < kcap; ++it) { pstep(ja, ona, pbeg, true); pstep(jb, onb, pend2, false); }
        if (ovf) atomicExch(&counters[CTR_OVERFLOW], 1);
        ja = ona ? ja : -1; jb = onb ? jb : -1;
        {
            const bool left = ona | onb;
            if (__any(left)) {
                const bool want = left && r3 < 0;
                const unsigned long long wb = __ballot(want);
                int slot = INT_MAX;
                if (wb) {
                    const int firstl = __ffsll((long long)wb) - 1;
                    int lbase = 0;
                    if (lane == firstl) lbase = atomicAdd(&l_nq, __popcll(wb));
                    lbase = __builtin_amdgcn_readlane(lbase, firstl);
                    if (want) slot = lbase + lane_rank(wb);
                }
                if (slot < qcap) {
                    lw[qbase + 2 * slot] = make_int2(h, me.x); lw[qbase + 2 * slot + 1] = make_int2(me.y, ja);
                    lx[qbase + 2 * slot] = make_int2(jb, r0); lx[qbase + 2 * slot + 1] = make_int2(r1, r2);
                    act = false;
                } else if (left) {
                    // (no room in the queue, or four components already: finish in place)
                    ja = walk(ja, pbeg, true, INT_MAX);
                    jb = walk(jb, pend2, false, INT_MAX);
                }
            }
        }
        finish(act, h, r0, r1, r2, r3);
    }
    __syncthreads();
    // ---- pass 1: the queued walkers, BQ_G lanes each ----
    const int nq = min(l_nq, qcap);
    const int gl = lane & (BQ_G - 1), gsh = lane & ~(BQ_G - 1);
    constexpr unsigned GM = BQ_G >= 32 ? 0xffffffffu : ((1u << (BQ_G & 31)) - 1u);
#pragma unroll 1
    for (int e0 = 0; e0 < nq; e0 += 256 / BQ_G) {
        const int e = e0 + ((int)threadIdx.x / BQ_G);
        const bool live = e < nq;
        int h = 0, ja = -1, jb = -1, r0 = -1, r1 = -1, r2 = -1, r3 = -1;
        int2 me = make_int2(0, 0);
        if (live) {
            const int2 e0v = lw[qbase + 2 * e], e1v = lw[qbase + 2 * e + 1], e2v = lx[qbase + 2 * e], e3v = lx[qbase + 2 * e + 1];
            h = e0v.x; me = make_int2(e0v.y, e1v.x); ja = e1v.y; jb = e2v.x; r0 = e2v.y; r1 = e3v.x; r2 = e3v.y;
        }
        const int qhi = me.x + g.eps;
        const int pbeg = me.y & ~(g.peps - 1), pend2 = pbeg + 2 * g.peps;
        const int plo = me.y - g.peps, phi = me.y + g.peps;
        // the group's lanes hold the same walker: one candidate per lane and step, the first that ends the window or lies within
        // reach decides (every lane of the group takes the same decision)
        auto gwalk = [&](int j, int pcap, bool below) {
            while (true) {
                const bool go = j >= 0 && j < C;
                if (!__any(go)) break;
                const int idx = j + gl;
                int2 p = make_int2(INT_MAX, INT_MAX);
                if (go && idx < C) { if (idx >= clo && idx < chi) p = lw[idx - clo]; else p = cpair[idx]; }
                const bool inw = (p.y < pcap) & (p.x <= qhi);
                const bool hit = inw & (below ? p.y >= plo : p.y <= phi);
                const unsigned gstop = (unsigned)(__ballot(go && (!inw | hit)) >> gsh) & GM;
                const unsigned ghit = (unsigned)(__ballot(go && hit) >> gsh) & GM;
                if (go) {
                    if (!gstop) j += BQ_G;
                    else {
                        const int f = __ffs((int)gstop) - 1;
                        if (!((ghit >> f) & 1u)) j = -1;
                        else {
                            const int jj = j + f;
                            int2 x;
                            if (jj >= clo && jj < chi) x = lx[jj - clo]; else x = make_int2(croot[jj], cskip[jj]);
                            see4(x.x, r0, r1, r2, r3, counters);
                            j = x.y;
                        }
                    }
                }
            }
        };
        gwalk(ja, pbeg, true);
        gwalk(jb, pend2, false);
        finish(live && gl == 0, h, r0, r1, r2, r3);
    }
}

// ---- variant 2 release rule (cDBSCAN2.py:180-183): the records of the contested walkers --------------------------
// The contested walkers that k_border_* listed (walkers of components that are not live on their cores alone), 16 lanes each.
// Only one whose first-come owner is UNCERTAIN can change hands (k_resolve_release walks a record's components in key order
// and a live one ends the walk): most listed walkers leave after that test.  The others collect their adjacent components:
// four walks over the core array (own strip downwards / upwards, one strip below, one above), 16 cores per round, every
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
    constexpr int EG = 16;                                // lanes per walker
    const int gl = lane & (EG - 1), gsh = lane & ~(EG - 1), grp = lane / EG;
    for (int k0 = 0; k0 < nlist; k0 += 64 * nwaves) {
        const int kl = k0 + lane * nwaves + wave;
        const int ci = kl < nlist ? clist[kl] : -1;
        const int co = ci >= 0 ? wowner[ci] : -1;
        unsigned long long todo = __ballot(ci >= 0 && state[owner_root(co) >= 0 ? owner_root(co) : 0] == ST_UNKNOWN && co >= 0);
        while (todo) {
            // the wave's next 64 / EG walkers, one per group of EG lanes (every lane of a group holds the same walker and takes the
            // same decisions; the kernel is a handful of dependent round trips per walker -- four of them side by side)
            int src = -1;
#pragma unroll
            for (int gi = 0; gi < 64 / EG; ++gi) {
                if (todo) { if (grp == gi) src = __ffsll((long long)todo) - 1; todo &= todo - 1; }
            }
            const bool act = src >= 0;
            const int h = __shfl(ci, act ? src : 0);
            const int2 me = act ? wpair[h] : make_int2(0, 0);
            const int pos = act ? wpos[h] : 0, enc = act ? wenc[h] : (int)LH_NONE;
            const int qlo = me.x - g.eps, qhi = me.x + g.eps;
            const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
            const int plo = me.y - g.peps, phi = me.y + g.peps;
            int c1 = 0, ca = C, cb = C;
            if (act) {
                c1 = core_rank(cmask, cgrank, pos);
                if ((unsigned)enc != LH_NONE) {
                    ca = core_rank(cmask, cgrank, pos - (int)((unsigned)enc & 0xffffu));
                    cb = core_rank(cmask, cgrank, pos + (int)((unsigned)enc >> 16));
                } else {
                    const int s = me.y >> g.rbits;
                    ca = s > 0 ? lower_bound_pairs(cpair, cstrip[s - 1], cstrip[s], qlo) : C;
                    cb = lower_bound_pairs(cpair, cstrip[s + 1], cstrip[min(s + 2, g.S)], qlo);
                }
            }
            int rr[4] = {-1, -1, -1, -1};
            int kk[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
            int nr = 0;
            bool any_u = false, overflow = false;
            auto see = [&](int r) {                                          // (every lane of the group keeps the same list)
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
            // one round: EG cores of walk w -> true if the group's walk goes on behind them
            auto round = [&](int w, bool on, bool in, int2 p, int r) {
                const bool ok = on && in && more(w, p.x, p.y);
                const int rv = (ok && acc(w, p.y)) ? r : -1;
                unsigned pending = (unsigned)(__ballot(rv >= 0) >> gsh) & 0xffffu;
                while (__any(pending != 0u)) {
                    const int R = __shfl(rv, gsh + (pending ? __ffs((int)pending) - 1 : 0));
                    const unsigned same = (unsigned)(__ballot(pending != 0u && rv == R) >> gsh) & 0xffffu;
                    if (pending) { see(R); pending &= ~same; }
                }
                return on && ((unsigned)(__ballot(ok) >> gsh) & 0xffffu) == 0xffffu;
            };
            // the first rounds of all four walks in flight together
            int2 p0[4]; int r0[4]; bool in0[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int j = w == 0 ? start[w] - gl : start[w] + gl;
                in0[w] = act && j >= 0 && j < C;
                p0[w] = in0[w] ? cpair[j] : make_int2(0, 0); r0[w] = in0[w] ? croot[j] : -1;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                bool on = round(w, act, in0[w], p0[w], r0[w]);
                for (int rnd = 1; __any(on); ++rnd) {
                    const int j = w == 0 ? start[w] - gl - EG * rnd : start[w] + gl + EG * rnd;
                    const bool in = on && j >= 0 && j < C;
                    on = round(w, on, in, in ? cpair[j] : make_int2(0, 0), in ? croot[j] : -1);
                }
            }
            if (act && gl == 0) {
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
#ifndef LF_TPB
#define LF_TPB BIGTPB
#endif
#ifndef LF_CHUNKS
#define LF_CHUNKS 2           // (2048 items per workgroup: 8 -> 79 us, 4 -> 61, 2 -> 53, 1 -> 57 per chr1 run -- the flush of a workgroup's table is a tail nothing overlaps but other workgroups)
#endif
__global__ void __launch_bounds__(LF_TPB)
k_final_lists(GridParams g, const int* __restrict__ lcnt, const int2* __restrict__ cpair, const int* __restrict__ cpos,
              const int* __restrict__ croot, const int2* __restrict__ wpair, const int* __restrict__ wpos,
              const int* __restrict__ wowner, const int* __restrict__ rlabel, const u32* __restrict__ srow,
              int* __restrict__ labels, int* __restrict__ llab, int* __restrict__ ldist, Table t,
              int2* __restrict__ pairs /* or null: (row, label) of every labelled item (compact, unordered; cl_wait copies pair_count of them out) */, int pairs_cap,
              int* __restrict__ pair_count)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int C = lcnt[0], L = C + lcnt[1];
    int idx[LF_CHUNKS], own[LF_CHUNKS], lab[LF_CHUNKS], x[LF_CHUNKS], y[LF_CHUNKS], pos[LF_CHUNKS];
    int2 pr[LF_CHUNKS];
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) {
        idx[ch] = (blockIdx.x * LF_CHUNKS + ch) * LF_TPB + threadIdx.x;
        const int k = idx[ch];
        own[ch] = -1; pos[ch] = 0; pr[ch] = make_int2(0, 0);
        if (k < C) { own[ch] = croot[k]; pr[ch] = cpair[k]; pos[ch] = (labels || pairs) ? cpos[k] : 0; }
        else if (k < L) { own[ch] = owner_root(wowner[k - C]); pr[ch] = wpair[k - C]; pos[ch] = (labels || pairs) ? wpos[k - C] : 0; }
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
    if (pairs) {
        // the reference's `.labels` holds clustered points only (cDBSCAN2.py:186-191): (row, label) pairs of the labelled items,
        // compacted -- ONE counter bump per workgroup (a bump per wave put 120 k same-address atomics with a return value into a
        // chr1 run), 8 contiguous bytes per lane.  (Written straight into page-locked host memory the kernel ran at 20 GB/s of
        // PCIe and held the compute stream.)
        __shared__ int l_pc[LF_CHUNKS * (LF_TPB / 64)];
        __shared__ int l_pbase;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned long long bals[LF_CHUNKS];
#pragma unroll
        for (int ch = 0; ch < LF_CHUNKS; ++ch) {
            bals[ch] = __ballot(lab[ch] >= 0);
            if (lane == 0) l_pc[ch * (LF_TPB / 64) + wv] = __popcll(bals[ch]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int k = 0; k < LF_CHUNKS * (LF_TPB / 64); ++k) { const int v = l_pc[k]; l_pc[k] = tot; tot += v; }      // (exclusive, in place)
            l_pbase = tot ? atomicAdd(pair_count, tot) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < LF_CHUNKS; ++ch) {
            if (lab[ch] < 0) continue;
            const int at = l_pbase + l_pc[ch * (LF_TPB / 64) + wv] + lane_rank(bals[ch]);
            if (at < pairs_cap) pairs[at] = make_int2((int)srow[pos[ch]], lab[ch]);
        }
    }
    if (L_ABL(1 << 10)) return;                             // (ablation: no cluster table)
#pragma unroll
    for (int ch = 0; ch < LF_CHUNKS; ++ch) table_accumulate(t, h, lab[ch], x[ch], y[ch]);
    if (L_ABL(1 << 9)) return;                              // (ablation: the table stays in LDS)
    table_flush(t, h);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ListBufs {
    unsigned long long *cmask, *wmask, *hmask;
    int *cgloc, *wgloc, *cgrank, *wgrank;
    int *bsum, *sup, *lcnt;
    int nsup2;                                           // ints of the superblock sums (both halves)
};
static int list_bufs(cl_chrom* c, const GridParams& g, int nm, ListBufs* b)
{
    const size_t n = (size_t)c->n;
    const size_t ngrp_cap = n / 64 + LG + 4;             // groups of any run of the handle (+ the one behind the last tile)
    const size_t nblk_cap = n / LT + 4;
    int rc;
    if ((rc = c->l_mask.ensure(3 * ngrp_cap * 8)) || (rc = c->l_rank.ensure(4 * ngrp_cap * 4)) ||
        (rc = c->l_blk.ensure((2 * (nblk_cap + 1) + 2 * (nblk_cap / 16 + 4) + 16) * 4)) || (rc = c->l_cstrip.ensure(((size_t)g.S + 4) * 4)) ||
        (rc = c->l_wpos.ensure(n * 4)) || (rc = c->l_wenc.ensure(n * 4)) || (rc = c->l_dist.ensure(n * 4)) ||
        (c->traversal < 3 && (rc = c->l_aux.ensure(n * 4)))) return rc;
    b->cmask = c->l_mask.as<unsigned long long>(); b->wmask = b->cmask + ngrp_cap; b->hmask = b->wmask + ngrp_cap;
    b->cgloc = c->l_rank.as<int>(); b->wgloc = b->cgloc + ngrp_cap; b->cgrank = b->wgloc + ngrp_cap; b->wgrank = b->cgrank + ngrp_cap;
    // l_blk: {C, W} | superblock sums (both halves; zero between runs: k_chain_c puts them back) | the tiles' sums
    const int nblk = nblocks(nm, LT);
    const size_t nsup_cap = nblk_cap / 16 + 4;          // (level 3: two int halves of 64-tile superblocks; level 4: one 64-bit word per 16 tiles)
    if (c->l_blk.fresh) {
        HIP_TRY(hipMemsetAsync(c->l_blk.p, 0, c->l_blk.bytes, c->stream));
        c->l_blk.fresh = false;
    }
    b->lcnt = c->l_blk.as<int>();
    b->sup = b->lcnt + 8;
    b->bsum = b->sup + 2 * nsup_cap;
    b->nsup2 = 2 * ((nblk + 63) / 64 + 1);
    return CL_OK;
}

static void launch_classify(cl_chrom* c, const GridParams& g, int nblk, const ListBufs& b)
{
#define LA_ARGS g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws, (const u32*)c->srow, b.cmask, b.wmask, b.hmask, b.cgloc, b.wgloc, b.bsum, b.sup, c->cellfirst.as<int>()
    if (g.variant == CL_VARIANT_CDBSCAN2) hipLaunchKernelGGL(k_classify<true>, dim3(nblk), dim3(256), 0, c->stream, LA_ARGS);
    else hipLaunchKernelGGL(k_classify<false>, dim3(nblk), dim3(256), 0, c->stream, LA_ARGS);
#undef LA_ARGS
}
static void launch_make_lists(cl_chrom* c, const GridParams& g, int nblk, const ListBufs& b, const ListRun& L)
{
#define LM_ARGS g, (const int*)c->w_sv, (const int*)c->w_sa, (const int*)c->w_strip, c->ws, (const u32*)c->srow, (const unsigned long long*)b.cmask,                   \
                (const unsigned long long*)b.wmask, (const unsigned long long*)b.hmask, (const int*)b.cgloc, (const int*)b.wgloc, (const int*)b.bsum, (const int*)b.sup,  \
                (const int*)c->cellfirst.as<int>(), b.cgrank, b.wgrank, L.cpair, L.cpos, L.ckey, L.wpair, L.wpos, L.wenc, b.lcnt
    if (g.variant == CL_VARIANT_CDBSCAN2) hipLaunchKernelGGL(k_make_lists<true>, dim3(nblk), dim3(256), 0, c->stream, LM_ARGS);
    else hipLaunchKernelGGL(k_make_lists<false>, dim3(nblk), dim3(256), 0, c->stream, LM_ARGS);
#undef LM_ARGS
}
static void list_views(cl_chrom* c, const ListBufs& b, ListRun* L)
{
    L->lcnt = b.lcnt; L->cmask = b.cmask; L->wmask = b.wmask; L->cgrank = b.cgrank; L->wgrank = b.wgrank;
    L->cpair = c->keys_in.as<int2>(); L->wpair = c->keys_out.as<int2>();     // (the sort buffers are dead behind the layout)
    L->cpos = c->head.as<int>(); L->ckey = c->hi.as<int>();
    L->wpos = c->l_wpos.as<int>(); L->wenc = c->l_wenc.as<int>(); L->cstrip = c->l_cstrip.as<int>();
}

static inline void fuse_knob(cl_chrom* c)
{
#ifdef CLOOPS_DEVEL
    const char* e = getenv("CLOOPS_FUSE");               // 0: k_chain_c + k_union_c<.., false> (the chains as a kernel of their own)
    if (e) c->fuse_chains = atoi(e) != 0;
#else
    (void)c;
#endif
}

int lists_build(cl_chrom* c, const GridParams& g, int nm, ListRun* out)
{
    ListBufs b;
    int rc = list_bufs(c, g, nm, &b);
    if (rc) return rc;
    const int nblk = nblocks(nm, LT);
    ListRun L{};
    list_views(c, b, &L);
    // the superblock sums are zero between runs (k_chain_c puts them back); a run that failed half-way leaves the flag up
    if (c->l_sup_dirty) HIP_TRY(hipMemsetAsync(b.sup, 0, (size_t)b.nsup2 * 4, c->stream));
    c->l_sup_dirty = true;
    launch_classify(c, g, nblk, b);
    launch_make_lists(c, g, nblk, b, L);
    // chains (the core count is only known on the device: the grids are sized by the PETs of the run)
    fuse_knob(c);
    if (c->fuse_chains)
        hipLaunchKernelGGL(k_prep_c, dim3(nblocks(std::max(std::max(nm, g.S + 2), b.nsup2))), dim3(TPB), 0, c->stream, g, (const int*)L.lcnt, (const int*)c->w_strip,
                           (const unsigned long long*)L.cmask, (const int*)L.cgrank, L.cstrip, b.sup, b.nsup2, c->parent.as<int>());
#ifdef CLOOPS_DEVEL
    else
    hipLaunchKernelGGL(k_chain_c, dim3(nblocks(nm, 256 * CH_PER)), dim3(256), 0, c->stream, g, L.lcnt, (const int2*)L.cpair, c->chainflag.as<int>(),
                       c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                       c->cellfirst.as<int>() /* cskip */, (const int*)c->w_strip, L.cmask, L.cgrank, L.cstrip, b.sup, b.nsup2, (int*)nullptr);
#else
    { }
#endif
    L.npos = nm; L.pstrip = c->w_strip;
    *out = L;
    HIP_TRY(hipGetLastError());
    c->l_sup_dirty = false;
    return CL_OK;
}

int lists_base_keys(cl_chrom* c, const GridParams& g)
{
    const int n = (int)c->n;
    int rc = c->bkey.ensure((size_t)n * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(k_base_keys, dim3(nblocks(n, LT)), dim3(256), 0, c->stream, g, n, (const int*)(c->bq.as<int>() + SORT_PAD),
                       (const int*)(c->bsp.as<int>() + SORT_PAD), (const u32*)c->brow.as<u32>(), c->bkey.as<int>());
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

int lists_build_base(cl_chrom* c, const GridParams& g, int nm, ListRun* out)
{
    fuse_knob(c);
    const int n = (int)c->n;
    ListBufs b;
    int rc = list_bufs(c, g, n, &b);
    if (rc) return rc;
    const int nblk = nblocks(n, LT);
    const bool v2 = g.variant == CL_VARIANT_CDBSCAN2;
    ListRun L{};
    list_views(c, b, &L);
    L.npos = n; L.pstrip = c->bstrip.as<int>();
    unsigned long long* sup64 = (unsigned long long*)b.sup;
    const int nsup_ints = 2 * (nblk / 16 + 2);
    if (c->l_sup_dirty) HIP_TRY(hipMemsetAsync(b.sup, 0, (size_t)std::max(nsup_ints, b.nsup2) * 4, c->stream));
    c->l_sup_dirty = true;
    const int* bq = c->bq.as<int>() + SORT_PAD;
    const int* bsp = c->bsp.as<int>() + SORT_PAD;
    const u32* brow = c->brow.as<u32>();
    c->srow = c->brow.as<u32>();                         // positions are base positions: their input rows
    const int* words = c->rc_cnt.as<int>();              // the handle's count cache, base-position space
    const int* band = c->cnt.as<int>();                  // fresh words of the cut band (k_band), base positions
    const int thr = c->l4_cut ? g.cut - g.V0 : INT_MIN;
    const int bandq = c->ws.bandq;
#define LCQ_ARGS g, n, thr, bandq, bq, words, band, b.cmask, b.wmask, b.cgloc, b.wgloc, b.bsum, sup64
    if (!c->l4_cut) hipLaunchKernelGGL((k_classify_q<false, false>), dim3(nblk), dim3(256), 0, c->stream, LCQ_ARGS);
    else if (c->l4_band) hipLaunchKernelGGL((k_classify_q<true, true>), dim3(nblk), dim3(256), 0, c->stream, LCQ_ARGS);
    else hipLaunchKernelGGL((k_classify_q<true, false>), dim3(nblk), dim3(256), 0, c->stream, LCQ_ARGS);
#undef LCQ_ARGS
#define LMQ_ARGS g, n, bandq, bq, bsp, brow, (const int*)c->bkey.as<int>(), (const int2*)c->l_fix.as<int2>(), words, band, (const unsigned long long*)b.cmask,                  \
                 (const unsigned long long*)b.wmask, (const int*)b.cgloc, (const int*)b.wgloc, (const int*)b.bsum, (const unsigned long long*)sup64, b.cgrank, b.wgrank, L.cpair,  \
                 c->run_rows ? L.cpos : (int*)nullptr, L.ckey, L.wpair, L.wpos, L.wenc, b.lcnt, c->fuse_chains ? c->parent.as<int>() : (int*)nullptr
    if (v2) {
        if (!c->l4_cut) hipLaunchKernelGGL((k_make_lists_q<true, false, false>), dim3(nblk), dim3(256), 0, c->stream, LMQ_ARGS);
        else if (c->l4_band) hipLaunchKernelGGL((k_make_lists_q<true, true, true>), dim3(nblk), dim3(256), 0, c->stream, LMQ_ARGS);
        else hipLaunchKernelGGL((k_make_lists_q<true, true, false>), dim3(nblk), dim3(256), 0, c->stream, LMQ_ARGS);
    } else {
        if (c->l4_band) hipLaunchKernelGGL((k_make_lists_q<false, true, true>), dim3(nblk), dim3(256), 0, c->stream, LMQ_ARGS);
        else hipLaunchKernelGGL((k_make_lists_q<false, false, false>), dim3(nblk), dim3(256), 0, c->stream, LMQ_ARGS);
    }
#undef LMQ_ARGS
    fuse_knob(c);
    if (c->fuse_chains)
        hipLaunchKernelGGL(k_prep_c, dim3(nblocks(std::max(g.S + 2, std::max(nsup_ints, b.nsup2)))), dim3(TPB), 0, c->stream, g, (const int*)L.lcnt,
                           (const int*)L.pstrip, (const unsigned long long*)L.cmask, (const int*)L.cgrank, L.cstrip, b.sup, std::max(nsup_ints, b.nsup2), (int*)nullptr);
#ifdef CLOOPS_DEVEL
    else
    hipLaunchKernelGGL(k_chain_c, dim3(nblocks(nm, 256 * CH_PER)), dim3(256), 0, c->stream, g, L.lcnt, (const int2*)L.cpair, c->chainflag.as<int>(),
                       c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                       c->cellfirst.as<int>() /* cskip */, L.pstrip, L.cmask, L.cgrank, L.cstrip, b.sup, std::max(nsup_ints, b.nsup2), v2 ? L.ckey : (int*)nullptr);
#else
    { }
#endif
    *out = L;
    HIP_TRY(hipGetLastError());
    c->l_sup_dirty = false;
    return CL_OK;
}

// level 4, first run of an eps under a cut: the words of the compact copy (c->cnt, by run position) to their base positions in the
// handle's cache, hints in base positions
int lists_words_to_base(cl_chrom* c, const GridParams& g)
{
    const int n = (int)c->n;
    LAUNCH(k_words_to_base, n, g, n, (const int*)(c->bsp.as<int>() + SORT_PAD), (const int4*)c->l_tab.as<int4>(), (const int*)c->rc_poff.as<int>(),
           (const int*)c->cnt.as<int>(), c->rc_cnt.as<int>());
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

// croot: c->root (levels >= 2) or, while the tile kernels still read roots by position (level 1), the distance-list buffer
static int* croot_of(cl_chrom* c) { return c->run_level >= 2 ? c->root.as<int>() : c->l_aux.as<int>(); }

int lists_union_flatten(cl_chrom* c, const GridParams& g, int nm, const ListRun& L)
{
    constexpr int UNT = 512;
    const int nt = nblocks(nm, UNT);
    // the union walk looks one strip back, i.e. about one strip's cores in front of the core: the halo follows the mean strip population
    const HeadReset hr{c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>()};
    int* prune = (c->run_level >= 4 && g.variant == CL_VARIANT_CDBSCAN2) ? L.ckey : (int*)nullptr;
#define LU_ARGS g, nt, L.lcnt, (const int2*)L.cpair, c->chainflag.as<int>(), (const int*)L.cstrip, c->cellfirst.as<int>() /* cskip: the cell minima are in the keys */, \
                c->parent.as<int>(), hr, prune, c->ulist.as<int>() /* the overflow list: free until k_mark_uncertain_l */, c->counters.as<int>()
    if (c->fuse_chains) {
        if ((long long)c->n > 80LL * g.S) hipLaunchKernelGGL((k_union_c<UNT, 512, true>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LU_ARGS);
        else hipLaunchKernelGGL((k_union_c<UNT, 128, true>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LU_ARGS);
        hipLaunchKernelGGL(k_union_overflow, dim3(1024), dim3(256), 0, c->stream, g, (const int*)c->counters.as<int>(), (const int*)c->ulist.as<int>(), (const int2*)L.cpair,
                           (const int*)c->chainflag.as<int>(), (const int*)L.cstrip, (const int*)c->cellfirst.as<int>(), c->parent.as<int>());
    }
#ifdef CLOOPS_DEVEL
    else {
        if ((long long)c->n > 80LL * g.S) hipLaunchKernelGGL((k_union_c<UNT, 512, false>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LU_ARGS);
        else hipLaunchKernelGGL((k_union_c<UNT, 128, false>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LU_ARGS);
    }
#endif
#undef LU_ARGS
    hipLaunchKernelGGL(k_flatten_c, dim3(nblocks(nm, FLC_TPB * FLC_PER)), dim3(FLC_TPB), 0, c->stream, L.lcnt, (const int*)c->chainflag.as<int>(),
                       (const int*)c->parent.as<int>(), (const int*)L.ckey, croot_of(c),
                       c->compkey.as<int>(), c->ncore.as<int>(), c->rootlist.as<int>(), c->counters.as<int>(), (int)g.dbg2);
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
    constexpr int BNT = 2048, BHC = 256, BNQ = 1024, BHQ = 254;      // (k_border_q: 1024-position tiles, six workgroups per CU -- 106 us against 110 for 2048 and 147 for 512 on the chr1 replay)
    (void)nm;
    int kcap = 2, bqg = 16;
    bool queued = true;
#ifdef CLOOPS_DEVEL
    { const char* e = getenv("CLOOPS_BCAP"); if (e) kcap = atoi(e); queued = kcap > 0; e = getenv("CLOOPS_BQG"); if (e) bqg = atoi(e); }
#endif
    const int nt = nblocks(L.npos, (g.variant == CL_VARIANT_CDBSCAN1 || !queued) ? BNT : BNQ);
#define LB_ARGS g, nt, L.npos, L.lcnt, L.cmask, L.cgrank, L.wgrank, (const int2*)L.cpair, (const int*)croot_of(c), (const int*)c->cellfirst.as<int>(),    \
                (const int*)L.ckey, (const int*)L.cstrip, (const int2*)L.wpair, (const int*)L.wpos, (const int*)L.wenc, (const int*)c->compkey.as<int>(),                    \
                (const int*)c->ncore.as<int>(), c->owner.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->chainflag.as<int>() /* clist: the chain ids are dead */,  \
                c->counters.as<int>()
    if (g.variant == CL_VARIANT_CDBSCAN1) hipLaunchKernelGGL((k_border_w<BNT, BHC, true>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LB_ARGS);
#ifdef CLOOPS_DEVEL
    else if (!queued) hipLaunchKernelGGL((k_border_w<BNT, BHC, false>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LB_ARGS);
#endif
#define LBQ_ARGS g, nt, L.npos, kcap, L.lcnt, L.cmask, L.cgrank, L.wgrank, (const int2*)L.cpair, (const int*)croot_of(c), (const int*)c->cellfirst.as<int>(),           \
                 (const int*)L.cstrip, (const int2*)L.wpair, (const int*)L.wpos, (const int*)L.wenc, (const int*)c->compkey.as<int>(), (const int*)c->ncore.as<int>(),        \
                 c->owner.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->chainflag.as<int>(), c->counters.as<int>()
#ifdef CLOOPS_DEVEL
    else if (bqg == 8) hipLaunchKernelGGL((k_border_q<BNQ, BHQ, 8>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LBQ_ARGS);
    else if (bqg == 32) hipLaunchKernelGGL((k_border_q<BNQ, BHQ, 32>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LBQ_ARGS);
#endif
    else hipLaunchKernelGGL((k_border_q<BNQ, BHQ, 16>), dim3(ltile_grid(nt)), dim3(256), 0, c->stream, LBQ_ARGS);
    (void)bqg;
#undef LBQ_ARGS
#undef LB_ARGS
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
    int* opos = c->l_aux.as<int>();
    HIP_TRY(hipMemsetAsync(opos, 0xFF, (size_t)nm * 4, c->stream));
    LAUNCH(k_scatter_by_pos, nm, L.lcnt, (const int*)L.cpos, (const int*)croot_of(c), opos);
    LAUNCH(k_scatter_by_pos, nm, L.lcnt + 1, (const int*)L.wpos, (const int*)c->owner.as<int>(), opos);
    return CL_OK;
}

int lists_final(cl_chrom* c, const GridParams& g, int nm, const ListRun& L, bool rows, int* pair_count)
{
    cl_chrom::Slot& sl = c->slot[c->cur];
    hipLaunchKernelGGL(k_final_lists, dim3(nblocks(nm, LF_TPB * LF_CHUNKS)), dim3(LF_TPB), 0, c->stream, g, L.lcnt, (const int2*)L.cpair, (const int*)L.cpos,
                       (const int*)croot_of(c), (const int2*)L.wpair, (const int*)L.wpos, (const int*)c->owner.as<int>(), (const int*)c->chainhead.as<int>(),
                       (const u32*)c->srow, (rows && !c->pairs_out) ? sl.labels.as<int>() : (int*)nullptr, sl.slab.as<int>(), c->l_dist.as<int>(), make_table(c),
                       c->pairs_out ? sl.pairs.as<int2>() : (int2*)nullptr, (int)std::min<long long>(c->pairs_cap, INT_MAX), pair_count);
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

// ------------------------------------------------------------------------------------------
// cl_cluster_rowmask_async: the labels of a run as ONE BIT PER ROW + the labels of the set rows in row order
// ------------------------------------------------------------------------------------------
// The reference's `.labels` holds the clustered points only (cDBSCAN2.py:186-191).  As (row, label) pairs that is 8 bytes per
// clustered PET over PCIe -- what bounds a label-inclusive sweep; as a bit per row and a label per set bit it is 4 bytes + 1/8
// byte per row, and the rows come back sorted.  Made from the run's row-aligned labels (-1 = not clustered) in two passes over
// 2048-row tiles: mask words + tile totals (+ the totals of 16-tile superblocks, one atomic nobody waits for), then the places.
#define RM_T 2048
__global__ void __launch_bounds__(256)
k_rowmask_count(int n, const int* __restrict__ labels, unsigned long long* __restrict__ mask, int* __restrict__ tsum,
                unsigned long long* __restrict__ sup)
{
    __shared__ int l_c[4];
    const int t0 = (int)blockIdx.x * RM_T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int lab[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int r = t0 + u * 256 + (int)threadIdx.x; lab[u] = r < n ? labels[r] : -1; }
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const unsigned long long bal = __ballot(lab[u] >= 0);
        if (lane == 0 && t0 + u * 256 + wv * 64 < n) mask[(t0 >> 6) + u * 4 + wv] = bal;
        cnt += __popcll(bal);
    }
    if (lane == 0) l_c[wv] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int tot = l_c[0] + l_c[1] + l_c[2] + l_c[3];
        tsum[blockIdx.x] = tot;
        if (tot) atomicAdd(&sup[blockIdx.x >> 4], (unsigned long long)tot);      // (a bump of ONE total by every tile cost 80 us: 8 000 atomics on one address)
    }
}
__global__ void __launch_bounds__(256)
k_rowmask_write(int n, const int* __restrict__ labels, const unsigned long long* __restrict__ mask, const int* __restrict__ tsum,
                const unsigned long long* __restrict__ sup, int* __restrict__ out, long long cap, int* __restrict__ total)
{
    __shared__ long long l_red[4];
    __shared__ int l_goff[RM_T / 64];
    const int blk = (int)blockIdx.x, t0 = blk * RM_T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int lab[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int r = t0 + u * 256 + (int)threadIdx.x; lab[u] = r < n ? labels[r] : -1; }
    // set rows in front of this tile: the superblocks in front of its own + the tiles of its own in front of it
    long long pc = 0;
    const int sb = blk >> 4;
    for (int k = threadIdx.x; k < sb; k += 256) pc += (long long)sup[k];
    if (threadIdx.x < 16) { const int k = sb * 16 + (int)threadIdx.x; if (k < blk) pc += tsum[k]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pc += __shfl_down(pc, o);
    if (lane == 0) l_red[wv] = pc;
    // the tile's 32 groups in row order: group k = u * 4 + wv holds rows t0 + 64 k ..
    if (threadIdx.x < 64) {
        const int k = lane & 31;
        const int gi = (t0 >> 6) + k;
        const int v = (lane < 32 && (long long)gi * 64 < n) ? __popcll(mask[gi]) : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) { const int t = __shfl_up(incl, d, 32); incl += k >= d ? t : 0; }
        if (lane < 32) l_goff[k] = incl - v;
    }
    __syncthreads();
    const long long base = l_red[0] + l_red[1] + l_red[2] + l_red[3];
    if (blk == (int)gridDim.x - 1 && threadIdx.x == 0) *total = (int)(base + tsum[blk]);      // the run's labelled PETs: header word 6
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const unsigned long long bal = __ballot(lab[u] >= 0);
        if (lab[u] >= 0) {
            const long long dst = base + l_goff[u * 4 + wv] + __popcll(bal & low_mask(lane));
            if (dst < cap) out[dst] = lab[u];
        }
    }
}
// the slot's `pairs` buffer in this form: mask words | labels (capacity) | tile sums | superblock sums
int lists_rowmask(cl_chrom* c, int* total)
{
    cl_chrom::Slot& sl = c->slot[c->cur];
    const long long n = c->n;
    const size_t nw = (size_t)((n + 63) / 64), ntile = (size_t)((n + RM_T - 1) / RM_T), nsup = ntile / 16 + 1;
    const size_t cap = (size_t)std::max<long long>(c->mask_cap, 1);
    const size_t o_lab = nw * 8, o_tsum = ((o_lab + cap * 4 + 255) / 256) * 256, o_sup = ((o_tsum + ntile * 4 + 255) / 256) * 256;
    int rc = sl.pairs.ensure(o_sup + nsup * 8 + 256);
    if (rc) return rc;
    char* p = (char*)sl.pairs.p;
    HIP_TRY(hipMemsetAsync(p + o_sup, 0, nsup * 8, c->stream));
    hipLaunchKernelGGL(k_rowmask_count, dim3((unsigned)ntile), dim3(256), 0, c->stream, (int)n, (const int*)sl.labels.as<int>(), (unsigned long long*)p,
                       (int*)(p + o_tsum), (unsigned long long*)(p + o_sup));
    hipLaunchKernelGGL(k_rowmask_write, dim3((unsigned)ntile), dim3(256), 0, c->stream, (int)n, (const int*)sl.labels.as<int>(), (const unsigned long long*)p,
                       (const int*)(p + o_tsum), (const unsigned long long*)(p + o_sup), (int*)(p + o_lab), (long long)c->mask_cap, total);
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

#ifdef CLOOPS_DEVEL
// developer build: one list kernel of the LAST run on its own, `reps` times between two events (its inputs are still in place;
// CLOOPS_DBG2 ablations apply).  which: 0 k_classify, 1 k_make_lists, 2 k_chain_c, 3 k_union_c, 4 k_border_w, 5 k_final_lists
extern "C" int cl_debug_time_lists(cl_chrom* c, int which, int reps, float* ms_out)
{
    if (!c || !ms_out || c->dbg_nm <= 0) return fail(CL_ERR_ARG, "cl_debug_time_lists: no rotated run on this handle");
    HIP_TRY(hipSetDevice(c->device));
    GridParams g = c->dbg_g;
    { const char* e = getenv("CLOOPS_DBG2"); g.dbg2 = e ? atoi(e) : 0; }
    const int nm = c->dbg_nm;
    ListBufs b;
    int rc = list_bufs(c, g, nm, &b);
    if (rc) return rc;
    const int nblk = nblocks(nm, LT);
    ListRun L{};
    list_views(c, b, &L);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) {
        if (which == 0) {
            // (the superblock sums must be zero on entry: k_chain_c does that in a run)
            HIP_TRY(hipMemsetAsync(b.sup, 0, (size_t)b.nsup2 * 4, c->stream));
            launch_classify(c, g, nblk, b);
        } else if (which == 1) {
            launch_make_lists(c, g, nblk, b, L);
        } else if (which == 6 || which == 7) {
            // level 4 (the last run must have been one with a cut that re-used counts): k_classify_q / k_make_lists_q
            const int n = (int)c->n;
            ListBufs bb;
            if ((rc = list_bufs(c, g, n, &bb))) return rc;
            const int nb4 = nblocks(n, LT);
            unsigned long long* sup64 = (unsigned long long*)bb.sup;
            const int* bq = c->bq.as<int>() + SORT_PAD; const int* bsp = c->bsp.as<int>() + SORT_PAD;
            const int thr = g.cut - g.V0, bandq = c->ws.bandq;
            if (which == 6) {
                HIP_TRY(hipMemsetAsync(bb.sup, 0, (size_t)2 * (nb4 / 16 + 2) * 4, c->stream));
                hipLaunchKernelGGL((k_classify_q<true, true>), dim3(nb4), dim3(256), 0, c->stream, g, n, thr, bandq, bq, (const int*)c->rc_cnt.as<int>(), (const int*)c->cnt.as<int>(),
                                   bb.cmask, bb.wmask, bb.cgloc, bb.wgloc, bb.bsum, sup64);
            } else {
                hipLaunchKernelGGL((k_make_lists_q<true, true, true>), dim3(nb4), dim3(256), 0, c->stream, g, n, bandq, bq, bsp, (const u32*)c->brow.as<u32>(), (const int*)c->bkey.as<int>(),
                                   (const int2*)c->l_fix.as<int2>(), (const int*)c->rc_cnt.as<int>(), (const int*)c->cnt.as<int>(), (const unsigned long long*)bb.cmask,
                                   (const unsigned long long*)bb.wmask, (const int*)bb.cgloc, (const int*)bb.wgloc, (const int*)bb.bsum, (const unsigned long long*)sup64, bb.cgrank, bb.wgrank,
                                   L.cpair, (int*)nullptr, L.ckey, L.wpair, L.wpos, L.wenc, bb.lcnt, (int*)nullptr);
            }
        } else if (which == 4 || which == 3 || which == 5) {
            // k_border_w / k_union_c + k_flatten_c / k_final_lists of the last run once more (their inputs are in place; the per-component
            // counters they add to are garbage afterwards)
            ListRun L2 = L;
            c->run_level = c->traversal >= 4 ? 4 : 3;
            if (c->run_level == 4) { ListBufs bb; if ((rc = list_bufs(c, g, (int)c->n, &bb))) return rc; list_views(c, bb, &L2); L2.npos = (int)c->n; L2.pstrip = c->bstrip.as<int>(); }
            else { L2.npos = nm; L2.pstrip = c->w_strip; }
            if (which == 4) { if ((rc = lists_border(c, g, nm, L2))) return rc; }
            else if (which == 3) {
                // (the chain kernel resets the forest and the per-head accumulators first -- timed with it)
                ListBufs bb; if ((rc = list_bufs(c, g, c->run_level == 4 ? (int)c->n : nm, &bb))) return rc;
                fuse_knob(c);
                if (c->fuse_chains)
                    hipLaunchKernelGGL(k_prep_c, dim3(nblocks(std::max(nm, g.S + 2))), dim3(TPB), 0, c->stream, g, (const int*)L2.lcnt, (const int*)L2.pstrip,
                                       (const unsigned long long*)L2.cmask, (const int*)L2.cgrank, L2.cstrip, bb.sup, 2, c->parent.as<int>());
#ifdef CLOOPS_DEVEL
                else
                hipLaunchKernelGGL(k_chain_c, dim3(nblocks(nm, 256 * CH_PER)), dim3(256), 0, c->stream, g, L2.lcnt, (const int2*)L2.cpair, c->chainflag.as<int>(),
                                   c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                                   c->cellfirst.as<int>(), L2.pstrip, L2.cmask, L2.cgrank, L2.cstrip, bb.sup, 2, (int*)nullptr);
#else
    { }
#endif
                HIP_TRY(hipMemsetAsync(c->counters.p, 0, 64, c->stream));
                if ((rc = lists_union_flatten(c, g, nm, L2))) return rc;
            } else { if ((rc = lists_final(c, g, nm, L2, false, c->counters.as<int>() + 40))) return rc; }
        } else return fail(CL_ERR_ARG, "cl_debug_time_lists: kernel not supported on its own");
    }
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    if (which == 1 || which == 7) HIP_TRY(hipMemsetAsync(b.sup, 0, c->l_blk.bytes - 64, c->stream));      // (what k_chain_c does in a run)
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / (float)std::max(1, reps);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return CL_OK;
}
#endif

// k_region.hip -- K2 of libcloops_hip.so: the region query (cDBSCAN.py:186-205 regionQuery / cDBSCAN2.py:304-334
// neighbour count) as LDS-tiled kernels over the sorted arrays of a run.  k_region_core is the HBM-roofline kernel
// of bench.py; DESIGN.md section 3 "K2 in detail".
#include "cl_common.h"

// One workgroup = a tile of 256 consecutive sorted PETs.  The tile plus a halo of K2_HALO PETs
// on both sides is staged in LDS as (q,p) pairs with coalesced loads.  With strips laid along
// v (a few tens of PETs per strip, uniformly) the three strips of a query sit next to each
// other in sorted order, so the whole region query runs out of LDS:
//   phase 1  own strip: every PET of the in-strip window [q-eps, q+eps] is a neighbour (the
//            strip coordinate differs by < eps), so that part of the count is an index
//            difference found by two branch-free 8-step searches -- no candidate is touched;
//   phase 2  strips s-1 / s+1, only for points not yet known to be core (DBSCAN needs
//            `count >= minPts`, not the count: EXACT = false saturates; cl_neighbor_counts()
//            instantiates EXACT = true).  Those points are first COMPACTED inside the
//            workgroup so that whole waves drop out instead of running at ~45 % lane use.
// Windows that leave the staged range (pile-ups of hundreds of PETs) continue in global memory.
// Workgroup b runs on XCD b % 8 (observed placement, used for speed only): each XCD is handed
// runs of K2_RUN consecutive tiles so that halos are re-read from its own L2.
template <bool EXACT>
__global__ void __launch_bounds__(K2_TPB)
k_region_count(GridParams g, int ntiles, int n, const int* __restrict__ sv, const int* __restrict__ sa,
               const int* __restrict__ strip_start, int* __restrict__ cnt)
{
    __shared__ int lq[K2_WIN], lp[K2_WIN];
    __shared__ int4 l_sb[K2_TPB];
    __shared__ short l_list[K2_TPB];
    __shared__ int l_wcount[K2_TPB / 64];
    const int xcd = blockIdx.x & 7, kseq = blockIdx.x >> 3;
    const int tile = ((kseq / K2_RUN) * 8 + xcd) * K2_RUN + (kseq % K2_RUN);
    const int t0 = tile * K2_TPB;
    if (tile >= ntiles) return;
    // M (PETs that passed the cut filter) lives on the device; the staging loads are predicated on the
    // host-known n instead (rows M..n-1 exist, they carry the sentinel strip), so that the load of M
    // overlaps the staging round trip instead of preceding it
    const int M = strip_start[g.S];
    const int base = t0 - K2_HALO;                 // global index of lq[0] / lp[0]
    {
        // all loads of the thread are issued before the first LDS store (a rolled loop waits for
        // its first round trip before it starts the second)
        static_assert(K2_WIN <= 2 * K2_TPB, "two staging slots per thread");
        const int k0 = threadIdx.x, k1 = threadIdx.x + K2_TPB;
        const int g0 = base + k0, g1 = base + k1;
        const bool in0 = g0 >= 0 && g0 < n, in1 = k1 < K2_WIN && g1 < n;
        int q0 = 0, p0v = 0, q1 = 0, p1v = 0;
        if (in0) { q0 = sv[g0]; p0v = sa[g0]; }
        if (in1) { q1 = sv[g1]; p1v = sa[g1]; }
        lq[k0] = q0; lp[k0] = p0v;
        if (k1 < K2_WIN) { lq[k1] = q1; lp[k1] = p1v; }
    }
    __syncthreads();
    if (t0 >= M) return;
    LdsSoA w; w.q = lq; w.p = lp; w.base = base;   // w[global sorted index] = (in-strip coord q, strip coord p)
    const int wbeg = max(base, 0), wend = min(base + K2_WIN, M);
    const int i = t0 + threadIdx.x;
    const bool valid = i < M;
#ifdef CLOOPS_DEVEL
    if (g.dbg & 32) { if (valid) cnt[i] = w[i].x + w[i].y; return; }        // developer knob: staging only
#endif
    // ---- phase 0: one-read core test ----------------------------------------------------------
    // If the minPts-1 next (or previous) PETs of the own strip are within eps in q, the point is
    // core: interiors of clusters are settled by one or two LDS reads, without any search.  "Same
    // strip" is tested on the staged p values, so phase 0 needs no strip bounds: the four bounds of
    // every PET are requested here (all loads in flight together, L2 hits) but only land in LDS after
    // the first compaction barrier -- their round trip overlaps phase 0 instead of preceding it.
    const int m1 = g.minPts - 1;
    const bool p0 = !EXACT && g.minPts >= 1 && m1 <= K2_SPAN;
    bool hard = false;
    int4 sbv = make_int4(0, 0, 0, 0);
    if (valid) {
        bool done = false;
        const int2 me = w[i];
        const int s = strip_of(g, me.y);
        sbv.y = strip_start[s]; sbv.z = strip_start[s + 1];
        sbv.x = strip_start[max(s - 1, 0)];
        sbv.w = strip_start[min(s + 2, g.S)];           // s + 1 == S: strip_start[S] == e
        if (p0) {
            const int jr = i + m1, jl = i - m1;
            if (jr < wend && w.qat(jr) - me.x <= g.eps && strip_of(g, w.p[jr - base]) == s) done = true;
            else if (jl >= wbeg && me.x - w.qat(jl) <= g.eps && strip_of(g, w.p[jl - base]) == s) done = true;
        }
        if (done) cnt[i] = g.minPts; else hard = true;
    }
#ifdef CLOOPS_DEVEL
    if (g.dbg & 64) { if (valid && hard) cnt[i] = 0; return; }              // developer knob: phase 0 only
#endif
    // ---- workgroup compaction: whole waves drop out of the search phases ------------------------
    const int total = block_compact_with<K2_TPB>(hard, l_list, l_wcount, [&]() { if (hard) l_sb[threadIdx.x] = sbv; });
    if ((int)threadIdx.x >= total) return;
    {
        const int tix = l_list[threadIdx.x];
        const int ii = t0 + tix;
        const int2 me = w[ii];
        const int qi = me.x, pi = me.y;
        const int qlo = sat_add(qi, -g.eps), qhi = sat_add(qi, g.eps);
        const int4 sb4 = l_sb[tix];
        const int tb = sb4.x, b = sb4.y, e = sb4.z, te = sb4.w;
        // ---- phase 1: own strip, index difference of two branch-free searches -------------------
        int lo, hi;
        if (p0) {
            // phase 0 failed on both sides, so the window ends before the (minPts-1)-th PET on either
            // side (or at the strip bounds): the searches run over at most minPts-1 positions and never
            // leave the staged range
            const int Ls = max(b, ii - m1 + 1), Rs = min(e, ii + m1);
            if (m1 <= 7) { lo = lds_lower_bound8<3>(w, Ls, ii + 1, qlo); hi = lds_upper_bound8<3>(w, ii + 1, Rs, qhi); }
            else if (m1 <= 31) { lo = lds_lower_bound8<5>(w, Ls, ii + 1, qlo); hi = lds_upper_bound8<5>(w, ii + 1, Rs, qhi); }
            else { lo = lds_lower_bound8<7>(w, Ls, ii + 1, qlo); hi = lds_upper_bound8<7>(w, ii + 1, Rs, qhi); }
        } else {
            // exact counts: most windows hold < 31 PETs per side: 5 steps; a window that reaches the 31st
            // position is searched again with 8 steps, one that leaves the staged span continues in global memory
            const int Ls = max(max(b, wbeg), ii - 31);
            lo = lds_lower_bound8<5>(w, Ls, ii + 1, qlo);
            if (lo == Ls && Ls > b) {
                const int L = max(max(b, wbeg), ii - K2_SPAN);
                lo = lds_lower_bound8<8>(w, L, Ls + 1, qlo);
                if (lo == L && L > b) lo = lower_bound_4(sv, b, L, qlo);
            }
            const int Rs = min(min(e, wend), ii + 32);
            hi = lds_upper_bound8<5>(w, ii + 1, Rs, qhi);
            if (hi == Rs && Rs < e) {
                const int R = min(min(e, wend), ii + 1 + K2_SPAN);
                hi = lds_upper_bound8<8>(w, Rs, R, qhi);
                if (hi == R && R < e) hi = lower_bound_4(sv, R, e, sat_add(qhi, 1));
            }
        }
        int c = hi - lo;
#ifdef CLOOPS_DEVEL
        if (g.dbg & 128) { cnt[ii] = c; return; }                             // developer knob: no neighbour strips
#endif
        // ---- phase 2: neighbour strips, only while not known to be core -------------------------
        if (EXACT || c < g.minPts) {
            // strip s-1 = [tb, b), strip s+1 = [e, te); an EMPTY strip counts as staged (the searches
            // return at once), so that a wave only leaves the common path for unstaged / very long strips
            const bool ldsA = tb >= wbeg && b - tb <= 255;
            const bool ldsB = te <= wend && te - e <= 255;
            if (ldsA && ldsB) {
                // both searches advance together (two independent LDS chains in flight); the number of
                // steps is chosen per WAVE (a per-lane choice makes most waves run every variant)
                const int longest = max(b - tb, te - e);
                int ja = tb, jb = e;
#define K2_PAIR_SEARCH(TOP)                                                                          \
                _Pragma("unroll") for (int step = TOP; step >= 1; step >>= 1) {                      \
                    const int ia = ja + step - 1, ib = jb + step - 1;                                \
                    const int va = w.qat(max(min(ia, b - 1), wbeg)), vb = w.qat(min(ib, te - 1));    \
                    ja = (ia < b && va < qlo) ? ja + step : ja;                                      \
                    jb = (ib < te && vb < qlo) ? jb + step : jb;                                     \
                }
                if (!__any(longest > 31)) { K2_PAIR_SEARCH(16) }
                else if (!__any(longest > 63)) { K2_PAIR_SEARCH(32) }
                else { K2_PAIR_SEARCH(128) }
#undef K2_PAIR_SEARCH
                // first four candidates of both strips, all loads in flight before the first compare
                int2 va[4], vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { va[k] = w[max(min(ja + k, b - 1), wbeg)]; vb[k] = w[min(jb + k, te - 1)]; }
                bool outA = false, outB = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ina = (ja + k < b) && (va[k].x <= qhi), inb = (jb + k < te) && (vb[k].x <= qhi);
                    outA |= !ina; outB |= !inb;
                    const int da = va[k].y - pi, db = vb[k].y - pi;
                    c += (ina && (da < 0 ? -da : da) <= g.peps) ? 1 : 0;
                    c += (inb && (db < 0 ? -db : db) <= g.peps) ? 1 : 0;
                }
                if (EXACT || c < g.minPts) {
                    if (!outA) c = k2_count_lds<EXACT, 4>(w, ja + 4, b, qhi, pi, g.peps, g.minPts, c);
                    if (!outB && (EXACT || c < g.minPts)) c = k2_count_lds<EXACT, 4>(w, jb + 4, te, qhi, pi, g.peps, g.minPts, c);
                }
            } else {
                if (tb < b) {
                    if (ldsA) {
                        const int j = (b - tb <= 63) ? lds_lower_bound8<6>(w, tb, b, qlo) : lds_lower_bound8<8>(w, tb, b, qlo);
                        c = k2_count_lds<EXACT, 4>(w, j, b, qhi, pi, g.peps, g.minPts, c);
                    } else {
                        const int j = lower_bound_4(sv, tb, b, qlo);
                        c = k2_count_glb<EXACT, 8>(sv, sa, j, b, qhi, pi, g.peps, g.minPts, c);
                    }
                }
                if ((EXACT || c < g.minPts) && e < te) {
                    if (ldsB) {
                        const int j = (te - e <= 63) ? lds_lower_bound8<6>(w, e, te, qlo) : lds_lower_bound8<8>(w, e, te, qlo);
                        c = k2_count_lds<EXACT, 4>(w, j, te, qhi, pi, g.peps, g.minPts, c);
                    } else {
                        const int j = lower_bound_4(sv, e, te, qlo);
                        c = k2_count_glb<EXACT, 8>(sv, sa, j, te, qhi, pi, g.peps, g.minPts, c);
                    }
                }
            }
        }
        cnt[ii] = c;
    }
}

// ------------------------------------------------------------------------------------------
// K2, clustering form (the roofline kernel).  DBSCAN does not need the neighbour count, only whether it
// reaches minPts, so this kernel decides "core or not" with as little work per PET as it can:
//   * a workgroup of 256 threads owns a tile of 256*U consecutive sorted PETs (U per thread) and stages the
//     tile plus a halo as two int arrays q[], sp[] (sp: GridParams) -- the halo and the two barriers are
//     amortised over U PETs per thread;
//   * the strip bounds come from a SLICE of the strip table staged next to the window (the slice starts at
//     the strip of the tile's first PET, which the sort phase left in tile_s0[]: one scalar load, then one
//     coalesced load per thread -- no per-PET global loads at all);
//   * phase 0, every PET: if the (minPts-1)-th next or previous PET lies in the same strip (one compare on
//     the staged sp: strips are aligned blocks of sp) within eps in q, the PET is core -- two LDS reads per side;
//   * the undecided PETs are appended to a list in LDS (one LDS atomic per wave) and handled by whole waves:
//     own strip = an index difference of two bounded branch-free searches; the strips s-1 / s+1 only while
//     the count is below minPts: both lower bounds by one paired search; where strips are long (dense data)
//     both upper bounds as well, so that "own + everything in both q windows < minPts" rejects a PET
//     without touching a candidate; candidates are tested 4 + 4 at a time (|dsp| <= peps is ONE compare
//     per side: a candidate one strip below can only be too low).
// Windows that leave the staged range fall back to global memory (pile-ups).  cl_neighbor_counts() (exact
// counts) and minPts outside 2..128 use k_region_count above.
// ------------------------------------------------------------------------------------------
// The sorted arrays sv / sa carry SORT_PAD sentinel entries in front of index 0 and behind index n-1 (left: q = 0,
// sp = INT_MIN; right: q = sp = INT_MAX -- "in no strip"), written once when the workspace is allocated: a tile
// window is staged with unpredicated 16-byte loads, no bounds logic at all.

#ifdef CLOOPS_DEVEL
#define K2_ABL(bit) do { if (g.dbg & (bit)) return; } while (0)      // developer ablation: stop after a phase (results invalid)
#else
#define K2_ABL(bit) do { } while (0)
#endif
#ifdef CLOOPS_DEVEL
// developer build: cycle stamps at the phase boundaries of k_region_core, kept in registers and stored once per wave
// at the very end (a store or atomic in the middle would be waited for by the next s_waitcnt and distort the phases)
__device__ unsigned int g_k2t[1 << 21];
#define K2T_INIT unsigned long long t_st[7]; t_st[0] = __builtin_readcyclecounter()
#define K2T(k) t_st[(k) + 1] = __builtin_readcyclecounter()
#define K2T_FLUSH do { if ((threadIdx.x & 63) == 0) { const unsigned w_ = (blockIdx.x * (K2F_TPB / 64) + (threadIdx.x >> 6)) & ((1u << 18) - 1u); \
    for (int k_ = 0; k_ < 6; ++k_) g_k2t[w_ * 8 + k_] = (unsigned)(t_st[k_ + 1] - t_st[k_]); g_k2t[w_ * 8 + 6] = ((unsigned)nh << 16) | (unsigned)n2; g_k2t[w_ * 8 + 7] = 1u; } } while (0)
#else
#define K2T(k) do { } while (0)
#define K2T_INIT do { } while (0)
#define K2T_FLUSH do { } while (0)
#endif

// The searches of k_region_core are written on PREDICATES OF THE STAGED PAIRS (q, sp), not on index bounds: sorted
// order is (strip, q) and strips are aligned blocks of sp, so "j is still before the window" is a monotone predicate
// of (q_j, sp_j) alone -- a probe is one 8-byte LDS read at an immediate offset, two or three compares and a select;
// no index compares, no clamps (probes may run a little past a strip: the window carries a halo and K2F_SLACK
// sentinel entries), no divergent branches.  first_true<K>(w, pos, pred): first index of [pos, pos + 2^K - 1] whose
// pair satisfies the monotone predicate (pos + 2^K - 1 if none does).
template <int K, typename P>
__device__ __forceinline__ int first_true(const int2* __restrict__ w, int pos, P&& pred)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const int2 v = w[pos + step - 1];
        pos = pred(v) ? pos : pos + step;
    }
    return pos;
}
// same with the probe index clamped to `last` (long brackets that may leave the LDS window)
template <int K, typename P>
__device__ __forceinline__ int first_true_clamped(const int2* __restrict__ w, int pos, int last, P&& pred)
{
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const int2 v = w[min(pos + step - 1, last)];
        pos = pred(v) ? pos : pos + step;
    }
    return pos;
}

// TAIL: the run has a cut, i.e. the sorted arrays end in a filtered tail whose tiles leave right after the scalar load of
// tile_s0 (before staging anything); without a cut every tile has work and the staging loads are issued BEFORE that load
// is waited for (its latency hides behind them).
// The word of a non-core PET (negative, cl_common.h "K2W"): its count in bits 24..30, in bits 0..11 / 12..23 the distance (in
// sorted positions) back to the start of its window in strip s-1 / forward to the one in strip s+1, all ones = no hints.
template <int U, int HALO, bool TAIL>
__global__ void __launch_bounds__(K2F_TPB)
k_region_core(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, const int* __restrict__ tile_s0, int* __restrict__ cnt)
{
    constexpr int TILE = K2F_TPB * U, WIN = TILE + 2 * HALO, NV = WIN / 4;
    constexpr int FULL = NV / K2F_TPB, REST = NV % K2F_TPB;              // int4 staging slots: FULL for every thread + a partial one
    constexpr int RUN = (2048 / TILE) > 0 ? (2048 / TILE) : 1;          // consecutive tiles per XCD (halo reuse in its L2)
    static_assert(HALO % 4 == 0 && HALO >= 128 && TILE + HALO + K2F_SLACK <= SORT_PAD && TILE % 256 == 0, "window shape");
    static_assert(WIN + K2F_SLACK < (int)K2H_MASK, "window offsets fit the hint fields");
    __shared__ __attribute__((aligned(16))) int2 lw[WIN + K2F_SLACK];   // (q, sp) pairs, window index = sorted index - (t0 - HALO)
    __shared__ int l_st[K2F_NS + 4];
    __shared__ unsigned int l_list[TILE];                                // undecided PETs, one region of 64*U entries per wave
    __shared__ unsigned char l_next[128];                                // smallest minPts of g.tmask above a count c (255: none)
    const int xcd = blockIdx.x & 7, kseq = blockIdx.x >> 3;
    const int tile = ((kseq / RUN) * 8 + xcd) * RUN + (kseq % RUN);
    if (tile >= ntiles) return;
    K2T_INIT;
    const int t0 = tile * TILE;
    int s0 = 0;                                         // strip of the tile's first PET; S = the tile lies in the filtered tail
    if (TAIL) { s0 = tile_s0[t0 >> 8]; if (s0 >= g.S) return; }
    const int M = strip_start[g.S];                     // PETs that passed the cut filter (device-side count)
    K2T(0);
    {
        // stage the window: unpredicated 16-byte loads (the arrays are padded with sentinels), every load of the
        // thread in flight before the first LDS store; pairs are interleaved on the way into LDS
        const int4* __restrict__ gq = reinterpret_cast<const int4*>(sv + (t0 - HALO));
        const int4* __restrict__ gp = reinterpret_cast<const int4*>(sa + (t0 - HALO));
        int4* l4 = reinterpret_cast<int4*>(lw);
        static_assert(REST == 0, "the window is a whole number of 16-byte slots per thread");
        int4 qv[FULL], pv[FULL];
#pragma unroll
        for (int u = 0; u < FULL; ++u) { qv[u] = gq[threadIdx.x + u * K2F_TPB]; pv[u] = gp[threadIdx.x + u * K2F_TPB]; }
        if (!TAIL) { s0 = tile_s0[t0 >> 8]; if (s0 >= g.S) return; }
        const int st = strip_start[min(max(s0 - 1 + (int)threadIdx.x, 0), g.S)];
#pragma unroll
        for (int u = 0; u < FULL; ++u) {
            const int k = (int)threadIdx.x + u * K2F_TPB;
            l4[2 * k] = make_int4(qv[u].x, pv[u].x, qv[u].y, pv[u].y);
            l4[2 * k + 1] = make_int4(qv[u].z, pv[u].z, qv[u].w, pv[u].w);
        }
        l_st[threadIdx.x] = st;
        if (threadIdx.x < 4) l_st[K2F_NS + threadIdx.x] = 0;
        if (threadIdx.x < 128) {
            // the served minPts values as a table: l_next[c] = the smallest one above c (bit t - 1 of the mask = minPts t)
            const int v = (int)threadIdx.x, wi = v >> 5;
            int nx = 255;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const u32 x = k == wi ? (g.tmask[k] & (~0u << (v & 31))) : (k > wi ? g.tmask[k] : 0u);
                if (x) nx = k * 32 + __ffs(x);
            }
            l_next[v] = (unsigned char)nx;
        }
        if (threadIdx.x < K2F_SLACK) lw[WIN + threadIdx.x] = make_int2(INT_MAX, INT_MAX);
    }
    K2T(1);
    __syncthreads();
    K2T(2);
    K2_ABL(32);
    const int m1 = g.minPts - 1;                        // 1 <= m1 <= 127 < HALO (the host guarantees it)
    const int eps = g.eps, peps = g.peps, minPts = g.minPts;
    const int nmask = ~(peps - 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned int* my_list = l_list + wv * (64 * U);
    int nh = 0;                                         // undecided PETs of this wave (wave-uniform)
    // ---- phase 0: one-read core test, U PETs per thread (all 3 * U LDS reads in flight before the first compare) ------
    int2 p_me[U], p_rr[U], p_ll[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int li = HALO + (int)threadIdx.x + u * K2F_TPB;
        p_me[u] = lw[li]; p_rr[u] = lw[li + m1]; p_ll[u] = lw[li - m1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int tix = (int)threadIdx.x + u * K2F_TPB;
        const int2 me = p_me[u], rr = p_rr[u], ll = p_ll[u];
        const int pbeg = me.y & nmask;
        const bool valid = (t0 + tix < M) & (me.x >= g.qmin);
        // the (minPts-1)-th next / previous PET is in the same strip and within eps in q (unsigned add: a sentinel q wraps harmlessly)
        const bool core = ((rr.y < pbeg + peps) & (rr.x <= (int)((unsigned)me.x + (unsigned)eps))) | ((ll.y >= pbeg) & (ll.x >= me.x - eps));
        if (valid & core) cnt[t0 + tix] = minPts;
        const bool hard = valid & !core;
        const unsigned long long bal = __ballot(hard);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (hard) my_list[nh + before] = (unsigned)tix;
        nh += __popcll(bal);
    }
    K2T(3);
    // the lists are per wave: a wave only reads what its own lanes wrote, and the LDS executes a wave's operations in order
    // -- no workgroup barrier, only "all my LDS writes have been issued" and a scheduling fence for the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    K2T(4);
    K2_ABL(64);
    // ---- phase 1: own strip.  Phase 0 failed on both sides, so the q window ends before the (minPts-1)-th PET on
    // either side (or at the strip ends): lo = first PET of the own strip with q >= qlo, hi = first PET behind the own
    // strip's PETs with q <= qhi, both inside +-(minPts-1) positions.  PETs still below minPts go on to phase 2
    // through a second list, written over the first one (a round appends at most as many entries as it has consumed).
    int n2 = 0;
    for (int h0 = 0; h0 < nh; h0 += 64) {
        const int h = h0 + lane;
        const bool act = h < nh;
        const int tix = act ? (int)my_list[h] : 0, li = HALO + tix;
        const int2 me = lw[li];
        const int qlo = me.x - eps, qhi = me.x + eps;   // q < 2^30, eps < 2^30: no overflow
        const int pbeg = me.y & nmask, pend = pbeg + peps;
        int lo, hi;
        auto inL = [&](int2 v) { return (v.y >= pbeg) & (v.x >= qlo); };            // monotone false -> true up to li
        auto outR = [&](int2 v) { return !((v.y < pend) & (v.x <= qhi)); };         // monotone false -> true from li + 1
        if (m1 <= 4) { lo = first_true<2>(lw, li - 3, inL); hi = first_true<2>(lw, li + 1, outR); }
        else if (m1 <= 8) { lo = first_true<3>(lw, li - 7, inL); hi = first_true<3>(lw, li + 1, outR); }
        else if (m1 <= 32) { lo = first_true<5>(lw, li - 31, inL); hi = first_true<5>(lw, li + 1, outR); }
        else if (m1 <= 64) { lo = first_true<6>(lw, li - 63, inL); hi = first_true<6>(lw, li + 1, outR); }
        else { lo = first_true<7>(lw, li - 127, inL); hi = first_true<7>(lw, li + 1, outR); }
        const int c = hi - lo;
        const bool need = act & (c < minPts);
        if (act & !need) cnt[t0 + tix] = c;
        const unsigned long long bal = __ballot(need);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (need) my_list[n2 + before] = (unsigned)tix | ((unsigned)c << 16);
        n2 += __popcll(bal);
    }
    K2_ABL(128);
    // ---- phase 2: neighbour strips s-1 = [tb, b) and s+1 = [e, te) -------------------------------------------
    const int off = HALO - t0;                          // window index = global sorted index + off
    const int wlo = max(t0 - HALO, 0) + off, whi = min(t0 - HALO + WIN, M) + off;      // staged valid range, window indices
    // A PET whose q windows in the neighbour strips hold enough PETs to reach minPts needs its candidates tested one by
    // one (|dp| <= eps) -- a loop whose trip count differs from lane to lane, and a wave pays the longest of its lanes.
    // Those PETs (one in seven on chr1 of the 200 M genome) go to a THIRD list and are counted by full waves afterwards
    // (phase 3) instead of stalling every round of phase 2: (tix | c << 16, ja | jb << 16) in the part of the wave's list
    // that phase 2 has already consumed.
    auto emit = [&](int tix, int li, int c, int hja, int hjb) {
        // a non-core PET leaves a NEGATIVE word (every consumer tests cnt >= minPts): K2H_ISOLATED if nothing can be within
        // eps of it, and where its windows in the neighbour strips start, relative to itself -- k_border walks them without
        // searching again (and without the strip table)
        int outv = c;
        if (c < minPts) {
            unsigned enc = 0x80000000u | ((unsigned)c << K2W_CSHIFT);
            enc |= (hja >= 0) ? ((unsigned)(li - hja) | ((unsigned)(hjb - li) << K2H_BITS)) : K2H_NONE;
            outv = (int)enc;
        }
        cnt[t0 + tix] = outv;
    };
    // depth of the upper-bound searches: a search that runs out of steps must leave an ub no served minPts lies above,
    // i.e. 2^K - 1 >= the widest gap between a count and the next served minPts (minPts - 1 for a one-off run)
    const int cap3 = g.tgap <= 31 ? 31 : (g.tgap <= 63 ? 63 : 127);
    auto count_candidates = [&](int c, int ja, int jb, int qhi, int pbeg, int pend2, int plo, int phi) {
        bool more = true;
        for (int j = ja; more & (c < minPts); j += 4) {
            int2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lw[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = (v[k].y < pbeg) & (v[k].x <= qhi);          // still a PET of strip s-1 inside the q window
                more &= in;
                c += (in & (v[k].y >= plo)) ? 1 : 0;                         // one strip below: sp can only be too low
            }
        }
        more = true;
        for (int j = jb; more & (c < minPts); j += 4) {
            int2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = lw[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = (v[k].y < pend2) & (v[k].x <= qhi);
                more &= in;
                c += (in & (v[k].y <= phi)) ? 1 : 0;                         // one strip above: only too high
            }
        }
        return c;
    };
    constexpr bool DEFER = HALO >= 512;                 // dense shapes
    int n3 = 0;
    for (int h = lane; h < n2; h += 64) {
        const unsigned ent = my_list[h];
        const int tix = (int)(ent & 0xffffu), li = HALO + tix;
        int c = (int)(ent >> 16);
        const int2 me = lw[li];
        const int qi = me.x, pi = me.y;
        const int qlo = qi - eps, qhi = qi + eps;
        const int pbeg = pi & nmask, pend2 = pbeg + 2 * peps;
        const int plo = pi - peps, phi = pi + peps;
        const int kk = (pi >> g.rbits) - s0;            // >= 0: the tile's PETs are in strips >= s0
        const int kc = min(kk, K2F_NS);                 // beyond the staged slice: dummy slots, fixed up below
        int tb = l_st[kc], b = l_st[kc + 1], e = l_st[kc + 2], te = l_st[kc + 3];
        if (kk + 3 >= K2F_NS) {
            const int s = kk + s0;
            tb = strip_start[max(s - 1, 0)]; b = strip_start[s]; e = strip_start[s + 1]; te = strip_start[min(s + 2, g.S)];
        }
        const int gtb0 = tb, gte0 = te;                 // global sorted indices, for the global-memory path
        tb += off; e += off; te += off;
        // A neighbour strip that sticks out of the staged range is CLIPPED to it when the staged part provably holds the
        // PET's q window (sorted by q: the first staged PET belongs to strip s-1 and lies below qlo / the last one to strip
        // s+1 above qhi) -- only a q window that itself leaves the staged range goes to global memory.  (Without this a few
        // per cent of the lanes -- strips of 150 .. 300 PETs against a 512-PET halo -- sent nearly every wave through
        // the global path as well.)
        bool okA = tb >= wlo, okB = te <= whi;
        if (DEFER && __any(!(okA & okB))) {             // (dense shapes; the sparse shape's short strips stay inside its halo)
            const int2 f = lw[wlo], l = lw[whi - 1];
            if (!okA) { okA = (f.y >= pbeg - peps) & (f.y < pbeg) & (f.x < qlo); tb = wlo; }
            if (!okB) { okB = (l.y >= pbeg + peps) & (l.y < pend2) & (l.x > qhi); te = whi; }
        }
        const int longest = max(b + off - tb, te - e);
#ifdef CLOOPS_DEVEL
        if (g.dbg & 1024) { cnt[t0 + tix] = c + longest; continue; }
        if (g.dbg & 256) { okA = okB = true; }
#endif
        int hja = -1, hjb = -1;                         // window starts in strips s-1 / s+1 (window indices), if found in LDS
        bool deferred = false;
        if (okA & okB) {
            auto inA = [&](int2 v) { return (v.y >= pbeg) | (v.x >= qlo); };    // from tb on: past the PETs of s-1 below qlo
            auto inB = [&](int2 v) { return (v.y >= pend2) | (v.x >= qlo); };   // from e on: past the PETs of s+1 below qlo
            if (!__any(longest > 31)) {
                // sparse data: 5-step searches, then the first two candidates of both strips at once
                const int ja = first_true<5>(lw, tb, inA), jb = first_true<5>(lw, e, inB);
                hja = ja; hjb = jb;
                int2 va[2], vb[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) { va[k] = lw[ja + k]; vb[k] = lw[jb + k]; }
                bool moreA = true, moreB = true;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const bool ia = (va[k].y < pbeg) & (va[k].x <= qhi), ib = (vb[k].y < pend2) & (vb[k].x <= qhi);
                    moreA &= ia; moreB &= ib;
                    c += (ia & (va[k].y >= plo)) ? 1 : 0;           // one strip below: sp can only be too low
                    c += (ib & (vb[k].y <= phi)) ? 1 : 0;           // one strip above: only too high
                }
                if (__any((c < minPts) & (moreA | moreB))) {
                    for (int j = ja + 2; moreA & (c < minPts); ++j) {
                        const int2 v = lw[j];
                        moreA = (v.y < pbeg) & (v.x <= qhi);
                        c += (moreA & (v.y >= plo)) ? 1 : 0;
                    }
                    for (int j = jb + 2; moreB & (c < minPts); ++j) {
                        const int2 v = lw[j];
                        moreB = (v.y < pend2) & (v.x <= qhi);
                        c += (moreB & (v.y <= phi)) ? 1 : 0;
                    }
                }
            } else {
                // dense data: both q windows [ja, ka), [jb, kb) first -- if even all of their PETs cannot lift the
                // count to minPts the PET is not core and no candidate is read
                auto outA = [&](int2 v) { return (v.y >= pbeg) | (v.x > qhi); };
                auto outB = [&](int2 v) { return (v.y >= pend2) | (v.x > qhi); };
                const int last = WIN + K2F_SLACK - 1;
                int ja, jb, ka, kb;
                // (which probes need their index clamped to the LDS array: a search from tb ends at or before the PET itself, so
                // its probes stay below li + 2^K - 1 -- inside the array for K <= FREEK (9 in the dense shapes); one from e may start at the
                // end of the window and is free only while 2^K - 1 <= K2F_SLACK; the upper-bound searches start inside the
                // window and reach at most cap3 <= K2F_SLACK - 1 entries further)
                constexpr int FREEK = HALO + K2F_SLACK > (1 << 9) - 3 ? 9 : (HALO + K2F_SLACK > (1 << 8) - 3 ? 8 : 7);      // li + 2^K - 2 < WIN + K2F_SLACK
                if (!__any(longest > 127)) { ja = first_true<7>(lw, tb, inA); jb = first_true<7>(lw, e, inB); }
                else if (!__any(longest > 255)) { ja = FREEK >= 8 ? first_true<8>(lw, tb, inA) : first_true_clamped<8>(lw, tb, last, inA); jb = first_true_clamped<8>(lw, e, last, inB); }
                else if (!__any(longest > 511)) { ja = FREEK >= 9 ? first_true<9>(lw, tb, inA) : first_true_clamped<9>(lw, tb, last, inA); jb = first_true_clamped<9>(lw, e, last, inB); }
                else if (!__any(longest > 1023)) { ja = first_true_clamped<10>(lw, tb, last, inA); jb = first_true_clamped<10>(lw, e, last, inB); }
                else { ja = first_true_clamped<12>(lw, tb, last, inA); jb = first_true_clamped<12>(lw, e, last, inB); }
                // the upper bounds only 2^K - 1 >= minPts - 1 positions deep: "do at least r = minPts - c more PETs follow in the
                // two windows" is all the rejection test needs (a search that runs out of steps reports 2^K - 1 >= r)
                static_assert(K2F_SLACK >= 128, "unclamped upper-bound searches");
                if (cap3 == 31) { ka = first_true<5>(lw, ja, outA); kb = first_true<5>(lw, jb, outB); }
                else if (cap3 == 63) { ka = first_true<6>(lw, ja, outA); kb = first_true<6>(lw, jb, outB); }
                else { ka = first_true<7>(lw, ja, outA); kb = first_true<7>(lw, jb, outB); }
                hja = ja; hjb = jb;
#ifdef CLOOPS_DEVEL
                if (g.dbg & 512) { cnt[t0 + tix] = c + ja + jb + ka + kb; continue; }
#endif
                // The count lies in [c, ub].  If none of the minPts values the words serve (g.tmask) falls into (c, ub], every
                // one of their tests reads the same from ub as from the count, and no candidate is read: not core at any of them
                // above c, core at every one up to c.  (A search that ran out of steps reports >= minPts - 1 positions: ub is
                // then >= minPts, which is in the set -- an ub that passes this test is the exact size of both windows, a true
                // upper bound of the count; <= 1 = isolated.)
                const int ub = c + (ka - ja) + (kb - jb);
                if (ub < (int)l_next[c]) c = ub;
                else {
                    // candidates to test: phase 3 (all lanes of this round have read their entries; what the round
                    // has consumed so far, 64 entries per round, is free -- an entry that would not fit is counted here)
                    const unsigned long long bal = __ballot(true);
                    const int slot = n3 + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    if (DEFER && 2 * slot + 1 < (h - lane) + 64) {
                        // (the sizes of the two q windows ride along, 7 bits each: a size below cap3 is exact -- the search did not
                        // run out of steps -- and phase 3 then walks the window by count, without testing where it ends)
                        my_list[2 * slot] = (ent & 0x7fffffu) | ((unsigned)(ka - ja) << 23) | ((unsigned)((kb - jb) & 3) << 30);
                        my_list[2 * slot + 1] = (unsigned)ja | ((unsigned)jb << 13) | ((unsigned)((kb - jb) >> 2) << 26);
                        deferred = true;
                    } else c = count_candidates(c, ja, jb, qhi, pbeg, pend2, plo, phi);
                }
            }
        } else {
            // a neighbour strip reaches outside the staged window (pile-up): global memory, sorted index space
            const int gtb = gtb0, ge = e - off, gte = gte0;
            if (gtb < b) {
                const int j = lower_bound_4(sv, gtb, b, qlo);
                c = k2_count_glb<false, 8>(sv, sa, j, b, qhi, pi, peps, minPts, c);
            }
            if (c < minPts && ge < gte) {
                const int j = lower_bound_4(sv, ge, gte, qlo);
                c = k2_count_glb<false, 8>(sv, sa, j, gte, qhi, pi, peps, minPts, c);
            }
        }
        // the number of deferred PETs of this round, seen by every lane that is still in the loop (wave-uniform count)
        if (DEFER) n3 += __popcll(__ballot(deferred));
        if (!deferred) emit(tix, li, c, hja, hjb);
    }
    // (the sparse shape never defers -- DEFER is a compile-time property of the tile shape: its strips are short, the few
    // candidate loops run in place and the tile ends here without the extra barrier)
    if (!DEFER) { K2T(5); K2T_FLUSH; return; }
    // every lane needs the final n3: lanes that left the loop early missed the later rounds
    n3 = __builtin_amdgcn_readfirstlane(dpp_reduce_wave(n3, OpMax()));
    // ---- phase 3: the candidates of the deferred PETs.  A wave defers a few tens of PETs of its 64 U -- a quarter of a
    // wave on chr1 of the 200 M genome, and the candidate loops run as long as the longest window of the wave -- so the four
    // lists are walked as ONE by the whole workgroup (one barrier): full waves first, the remainder in the last one.
    if (lane == 0) l_st[K2F_NS + wv] = n3;               // (the four dummy slots behind the strip-table slice: only phase 2 read them)
    __syncthreads();
    const int c0 = l_st[K2F_NS], c1 = c0 + l_st[K2F_NS + 1], c2 = c1 + l_st[K2F_NS + 2];
    int ntot = c2 + l_st[K2F_NS + 3];
#ifdef CLOOPS_DEVEL
    if (g.dbg & 2048) ntot = 0;
#endif
    for (int gi = (int)threadIdx.x; gi < ntot; gi += K2F_TPB) {
        const int w = (gi >= c0) + (gi >= c1) + (gi >= c2);
        const int h = gi - (w == 0 ? 0 : (w == 1 ? c0 : (w == 2 ? c1 : c2)));
        const unsigned int* wl = l_list + w * (64 * U);
        const unsigned ent = wl[2 * h], jj = wl[2 * h + 1];
        const int tix = (int)(ent & 0xffffu), li = HALO + tix;
        const int ja = (int)(jj & 0x1fffu), jb = (int)((jj >> 13) & 0x1fffu);
        const int na = (int)((ent >> 23) & 0x7fu), nb = (int)((ent >> 30) | ((jj >> 26) << 2));
        const int2 me = lw[li];
        const int pbeg = me.y & nmask;
        int c = (int)((ent >> 16) & 0x7fu);
        if (__any((na >= cap3) | (nb >= cap3)))
            c = count_candidates(c, ja, jb, me.x + eps, pbeg, pbeg + 2 * peps, me.y - peps, me.y + peps);
        else {
            // both windows are known exactly: [ja, ja + na) of strip s-1, [jb, jb + nb) of strip s+1 -- only the strip coordinate
            // is left to test (one strip below: sp can only be too low; one strip above: only too high)
            const int plo = me.y - peps, phi = me.y + peps;
            for (int j = 0; (j < na) & (c < minPts); j += 4) {
                int2 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lw[ja + j + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) c += ((j + k < na) & (v[k].y >= plo)) ? 1 : 0;
            }
            for (int j = 0; (j < nb) & (c < minPts); j += 4) {
                int2 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = lw[jb + j + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) c += ((j + k < nb) & (v[k].y <= phi)) ? 1 : 0;
            }
        }
        emit(tix, li, c, ja, jb);
    }
    K2T(5);
    K2T_FLUSH;
}

// ------------------------------------------------------------------------------------------
// K2, sorted-key form (round 6): the clustering form for the DENSE shapes, re-decomposed so that
//   (1) a search probe is ONE 8-byte LDS read at an immediate offset, one 64-bit compare and a select: the window is staged as
//       ONE sorted array of 64-bit keys  strip_rel << 52 | q << 24 | cum  (strip_rel = strip - (s0 - 2), clamped to 0 .. 4095;
//       q < 2^28: the host checks qtop + eps + 1 < 2^28) -- sorted order (strip, q) IS the order of the keys, so "first PET of
//       strip t with q >= x" is a plain lower bound of the key (t, x), whichever strip the probes run through;
//   (2) the candidates of the neighbour strips are not touched at all for most PETs: `cum` holds two RUNNING CLASS COUNTERS over
//       the window order (12 bits each: how many staged entries in front have r = p mod eps >= eps/3 resp. >= 2 eps/3), so for a q
//       window [ja, ka) of strip s-1 the number of its PETs in every third of the strip is a difference of two reads.  One strip
//       below a candidate is a neighbour iff r_j >= r_i: every PET of a HIGHER third is one, every PET of a lower third is not,
//       only the PETs of the query's own third are uncertain (one strip above: mirrored).  That gives count in [low, up] with
//       up - low = a third of the two windows; if no served minPts lies in (low, up] the word is settled (cDBSCAN.py:186-205
//       needs the count only against minPts; cDBSCAN2.py:333-334 likewise) -- on chr1 of the 200 M genome 3-9 % of the PETs are
//       left for candidate walks instead of 20-31 %;
//   (3) the in-strip remainders r are not staged: the walks of those few PETs read sp from global memory (L2 hits: the tile has
//       just streamed it).
// Everything else (tile / halo shapes, strip-table slice, phase 0, per-wave lists, deferred walks by the workgroup, the word
// and its hints, the global-memory continuation for pile-ups) is k_region_core's.
// ------------------------------------------------------------------------------------------
typedef unsigned long long u64k;
#ifdef CLOOPS_DEVEL
// developer build, CLOOPS_DBG bit 4096: how often the rare paths of k_region_keys run (tools/fuzz_k2_keys.py prints them): 0 launches, 1 PETs in
// phase 2, 2 beyond the strip_rel clamp, 3 through global memory, 4 clipped windows, 5 deferred walks, 6 walks in place, 7 capped windows
__device__ unsigned long long g_k2stat[8];
extern "C" void cl_debug_k2stats(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2stat), sizeof(g_k2stat));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2stat), z, sizeof(z));
}
#define K2STAT(slot, cond) do { if ((g.dbg & 4096) && (cond)) atomicAdd(&g_k2stat[slot], 1ull); } while (0)
#else
#define K2STAT(slot, cond) do { } while (0)
#endif
#define K2K_RELMAX 4095
#define K2K_HCAP 63           // the upper ends of the neighbour windows are searched 6 steps deep

// first index of [pos, pos + 2^K - 1] whose key is >= key (pos + 2^K - 1 if none of the probed ones is)
template <int K>
__device__ __forceinline__ int k2k_first_ge(const u64k* __restrict__ w, int pos, u64k key)
{
    // (the position runs in BYTES: a probe is ds_read_b64 at an immediate offset, v_cmp_lt_u64, v_add, v_cndmask -- no address shift)
    int p8 = pos * 8;
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const u64k v = *reinterpret_cast<const u64k*>(reinterpret_cast<const char*>(w) + p8 + (step - 1) * 8);
        p8 = v >= key ? p8 : p8 + step * 8;
    }
    return p8 >> 3;
}
template <int K>
__device__ __forceinline__ int k2k_first_ge_clamped(const u64k* __restrict__ w, int pos, int last, u64k key)
{
    int p8 = pos * 8;
    const int last8 = last * 8;
#pragma unroll
    for (int step = 1 << (K - 1); step >= 1; step >>= 1) {
        const u64k v = *reinterpret_cast<const u64k*>(reinterpret_cast<const char*>(w) + min(p8 + (step - 1) * 8, last8));
        p8 = v >= key ? p8 : p8 + step * 8;
    }
    return p8 >> 3;
}
// inclusive prefix sum over the 64 lanes of a wave on the DPP network
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

template <int U, int HALO, bool TAIL>
__global__ void __launch_bounds__(K2F_TPB, HALO >= 1024 ? 5 : 7)      // (the LDS holds seven / five workgroups per CU: the registers must too)
k_region_keys(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, const int* __restrict__ tile_s0, int* __restrict__ cnt)
{
    constexpr int TILE = K2F_TPB * U, WIN = TILE + 2 * HALO, NV = WIN / 4;
    constexpr int FULL = NV / K2F_TPB;                                   // int4 staging slots per thread
    constexpr int NSEG = FULL * (K2F_TPB / 64);                         // 256-entry segments of the window, one per (slot, wave)
    constexpr int RUN = (2048 / TILE) > 0 ? (2048 / TILE) : 1;
    static_assert(HALO % 4 == 0 && HALO >= 128 && TILE + HALO + K2F_SLACK <= SORT_PAD && TILE % 256 == 0 && NV % K2F_TPB == 0, "window shape");
    static_assert(WIN + K2F_SLACK < (int)K2H_MASK && WIN + K2F_SLACK < 4096 && K2F_SLACK > K2K_HCAP && NSEG <= 64 && TILE <= 1024, "field widths");
    __shared__ __attribute__((aligned(16))) u64k lw[WIN + K2F_SLACK];   // sorted keys, window index = sorted index - (t0 - HALO)
    __shared__ int l_st[K2F_NS + 4];
    __shared__ unsigned int l_list[TILE];
    __shared__ unsigned char l_next[128];
    unsigned int* l_seg = l_list;                                        // segment totals of the counter scan (read before the lists exist)
    const int xcd = blockIdx.x & 7, kseq = blockIdx.x >> 3;
    const int tile = ((kseq / RUN) * 8 + xcd) * RUN + (kseq % RUN);
    if (tile >= ntiles) return;
    const int t0 = tile * TILE;
    int s0 = 0;
    if (TAIL) { s0 = tile_s0[t0 >> 8]; if (s0 >= g.S) return; }
    const int M = strip_start[g.S];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int eps = g.eps, peps = g.peps, minPts = g.minPts;
    {
        const int4* __restrict__ gq = reinterpret_cast<const int4*>(sv + (t0 - HALO));
        const int4* __restrict__ gp = reinterpret_cast<const int4*>(sa + (t0 - HALO));
        int4 qv[FULL], pv[FULL];
#pragma unroll
        for (int u = 0; u < FULL; ++u) { qv[u] = gq[threadIdx.x + u * K2F_TPB]; pv[u] = gp[threadIdx.x + u * K2F_TPB]; }
        if (!TAIL) { s0 = tile_s0[t0 >> 8]; if (s0 >= g.S) return; }
        const int st = strip_start[min(max(s0 - 1 + (int)threadIdx.x, 0), g.S)];
        // class counters: running counts of r >= t1 (bits 0..11) and r >= t2 (bits 12..23) over the window order
        const int t1 = (eps + 2) / 3, t2 = (2 * eps + 2) / 3, rmask = peps - 1;
        unsigned cx[FULL][4], tot[FULL];
#pragma unroll
        for (int u = 0; u < FULL; ++u) {
            const int ps[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
            unsigned run = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = ps[k] & rmask;
                cx[u][k] = run;
                run += (r >= t1 ? 1u : 0u) + (r >= t2 ? 0x1000u : 0u);
            }
            tot[u] = wave_incl_scan(run);
            if (lane == 63) l_seg[u * (K2F_TPB / 64) + wv] = tot[u];
            tot[u] -= run;                              // exclusive inside the (slot, wave) segment
        }
        l_st[threadIdx.x] = st;
        if (threadIdx.x < 4) l_st[K2F_NS + threadIdx.x] = 0;
        if (threadIdx.x < 128) {
            const int v = (int)threadIdx.x, wi = v >> 5;
            int nx = 255;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const u32 x = k == wi ? (g.tmask[k] & (~0u << (v & 31))) : (k > wi ? g.tmask[k] : 0u);
                if (x) nx = k * 32 + __ffs(x);
            }
            l_next[v] = (unsigned char)nx;
        }
        __syncthreads();
        // the segments in front of mine: one read per lane, a wave scan, a read-lane per slot
        const unsigned sg = lane < NSEG ? l_seg[lane] : 0u;
        const unsigned sgi = wave_incl_scan(sg);
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)sgi, 63);
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const int sbase = s0 - 2;
        const bool nearM = t0 - HALO + WIN > M;        // rows behind M (a filtered tail) carry strip S: they are in no strip
        int4* l4 = reinterpret_cast<int4*>(lw);
#pragma unroll
        for (int u = 0; u < FULL; ++u) {
            const unsigned segbase = (unsigned)__builtin_amdgcn_readlane((int)(sgi - sg), u * (K2F_TPB / 64) + wvu);
            const unsigned base = segbase + tot[u];
            const int slot = (int)threadIdx.x + u * K2F_TPB;
            const int qs[4] = {qv[u].x, qv[u].y, qv[u].z, qv[u].w};
            const int ps[4] = {pv[u].x, pv[u].y, pv[u].z, pv[u].w};
            unsigned lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int rel = min(max((ps[k] >> g.rbits) - sbase, 0), K2K_RELMAX);
                if (nearM && t0 - HALO + 4 * slot + k >= M) rel = K2K_RELMAX;
                const unsigned q = min((unsigned)qs[k], 0x0fffffffu);
                lo[k] = (q << 24) | (base + cx[u][k]);
                hi[k] = ((unsigned)rel << 20) | (q >> 8);
            }
            l4[2 * slot] = make_int4((int)lo[0], (int)hi[0], (int)lo[1], (int)hi[1]);
            l4[2 * slot + 1] = make_int4((int)lo[2], (int)hi[2], (int)lo[3], (int)hi[3]);
        }
        if (threadIdx.x < K2F_SLACK) lw[WIN + threadIdx.x] = ((u64k)0xffffffffu << 32) | 0xff000000u | total;
    }
    __syncthreads();
    K2_ABL(32);
    K2STAT(0, threadIdx.x == 0 && blockIdx.x == 0);
    const int m1 = minPts - 1;
    const u64k E24 = (u64k)(unsigned)eps << 24, STRIP1 = 1ull << 52;
    const u64k KMASK = ~(u64k)0xffffffu;                // (strip_rel, q) of a key
    unsigned int* my_list = l_list + wv * (64 * U);
    int nh = 0;
    // ---- phase 0: the (minPts-1)-th next / previous PET is in the same strip and within eps in q -> core -------------------
    u64k p_me[U], p_rr[U], p_ll[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int li = HALO + (int)threadIdx.x + u * K2F_TPB;
        p_me[u] = lw[li]; p_rr[u] = lw[li + m1]; p_ll[u] = lw[li - m1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int tix = (int)threadIdx.x + u * K2F_TPB;
        const u64k me = p_me[u] & KMASK;
        const int qi = (int)((unsigned)(me >> 24) & 0x0fffffffu);
        const bool valid = (t0 + tix < M) & (qi >= g.qmin);
        // (a key of a later strip is above (rel, q + eps) whatever its q: q + eps < 2^28; one of an earlier strip, raised by eps, stays below)
        // (a PET at the strip_rel clamp shares its key's strip field with every strip behind it: no key test says "same strip" there)
        const bool core = ((p_rr[u] < me + E24 + (1ull << 24)) | (p_ll[u] + E24 >= me)) & ((unsigned)(me >> 32) < ((unsigned)K2K_RELMAX << 20));
        if (valid & core) cnt[t0 + tix] = minPts;
        const bool hard = valid & !core;
        const unsigned long long bal = __ballot(hard);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (hard) my_list[nh + before] = (unsigned)tix;
        nh += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    K2_ABL(64);
    // ---- phase 1: own strip = an index difference of two bounded searches ---------------------------------------------------
    int n2 = 0;
    for (int h0 = 0; h0 < nh; h0 += 64) {
        const int h = h0 + lane;
        const bool act = h < nh;
        const int tix = act ? (int)my_list[h] : 0, li = HALO + tix;
        const u64k me = lw[li] & KMASK;
        const unsigned qi = (unsigned)(me >> 24) & 0x0fffffffu;
        const u64k keyL = me - ((u64k)min(qi, (unsigned)eps) << 24), keyH = me + E24 + (1ull << 24);
        int lo, hi;
        if (m1 <= 4) { lo = k2k_first_ge<2>(lw, li - 3, keyL); hi = k2k_first_ge<2>(lw, li + 1, keyH); }
        else if (m1 <= 8) { lo = k2k_first_ge<3>(lw, li - 7, keyL); hi = k2k_first_ge<3>(lw, li + 1, keyH); }
        else if (m1 <= 32) { lo = k2k_first_ge<5>(lw, li - 31, keyL); hi = k2k_first_ge<5>(lw, li + 1, keyH); }
        else if (m1 <= 64) { lo = k2k_first_ge<6>(lw, li - 63, keyL); hi = k2k_first_ge<6>(lw, li + 1, keyH); }
        else { lo = k2k_first_ge<7>(lw, li - 127, keyL); hi = k2k_first_ge<7>(lw, li + 1, keyH); }
        int c = hi - lo;
        const bool clamped = act & ((unsigned)(me >> 32) >= ((unsigned)K2K_RELMAX << 20));
        if (__any(clamped)) {
            // at the strip_rel clamp (a window that spans more than 4 000 strips: a gap in the data) the own strip comes from the strip
            // table and global memory
            if (clamped) {
                const int st = sa[t0 + tix] >> g.rbits;
                const int b = strip_start[st], e = strip_start[st + 1];
                c = lower_bound_4(sv, b, e, (int)qi + eps + 1) - lower_bound_4(sv, b, e, (int)qi - eps);
            }
        }
        const bool need = act & (c < minPts);
        if (act & !need) cnt[t0 + tix] = c;
        const unsigned long long bal = __ballot(need);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (need) my_list[n2 + before] = (unsigned)tix | ((unsigned)c << 16);
        n2 += __popcll(bal);
    }
    K2_ABL(128);
    // ---- phase 2: the q windows [ja, ka) of strip s-1 and [jb, kb) of strip s+1, their class counts, the bracket -----------
    const int off = HALO - t0;
    const int wlo = max(t0 - HALO, 0) + off, whi = min(t0 - HALO + WIN, M) + off;
    const int last = WIN + K2F_SLACK - 1;
    auto emit = [&](int tix, int li, int c, int hja, int hjb) {
        int outv = c;
        if (c < minPts) {
            unsigned enc = 0x80000000u | ((unsigned)c << K2W_CSHIFT);
            enc |= (hja >= 0) ? ((unsigned)(li - hja) | ((unsigned)(hjb - li) << K2H_BITS)) : K2H_NONE;
            outv = (int)enc;
        }
        cnt[t0 + tix] = outv;
    };
    // the candidates of both windows one by one (their strip coordinates from global memory: window index + t0 - HALO)
    auto walk = [&](int c, int tix, int ja, int na, int jb, int nb) {
        const int pi = sa[t0 + tix], plo = pi - peps, phi = pi + peps;
        const int* __restrict__ ga = sa + (t0 - HALO) + ja;
        const int* __restrict__ gb = sa + (t0 - HALO) + jb;
        for (int j = 0; (j < na) & (c < minPts); j += 4) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ga[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) c += ((j + k < na) & (v[k] >= plo)) ? 1 : 0;       // one strip below: sp can only be too low
        }
        for (int j = 0; (j < nb) & (c < minPts); j += 4) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = gb[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) c += ((j + k < nb) & (v[k] <= phi)) ? 1 : 0;       // one strip above: only too high
        }
        return c;
    };
    int n3 = 0;
    for (int h = lane; h < n2; h += 64) {
        const unsigned ent = my_list[h];
        const int tix = (int)(ent & 0xffffu), li = HALO + tix;
        int c = (int)(ent >> 16);
        const u64k me = lw[li] & KMASK;
        const unsigned qi = (unsigned)(me >> 24) & 0x0fffffffu;
        const int rel = (int)(me >> 52);
        const bool far = rel >= K2K_RELMAX - 1;           // the clamp of strip_rel: the keys do not tell this PET's strips apart
        K2STAT(1, true); K2STAT(2, far);
        const int kk = far ? (sa[t0 + tix] >> g.rbits) - s0 : rel - 2;
        const int kc = min(kk, K2F_NS);
        int tb = l_st[kc], b = l_st[kc + 1], e = l_st[kc + 2], te = l_st[kc + 3];
        asm volatile("" : "+v"(tb), "+v"(b), "+v"(e), "+v"(te));      // (four LDS reads: keeps the compiler from merging them with the rare global loads below into FLAT loads)
        if (kk + 3 >= K2F_NS) {
            const int s = kk + s0;
            tb = strip_start[max(s - 1, 0)]; b = strip_start[s]; e = strip_start[s + 1]; te = strip_start[min(s + 2, g.S)];
        }
        const int gtb0 = tb, gte0 = te;
        tb += off; e += off; te += off;
        const u64k dq = (u64k)min(qi, (unsigned)eps) << 24;
        const u64k keyAL = me - STRIP1 - dq, keyAH = me - STRIP1 + E24 + (1ull << 24);
        const u64k keyBL = me + STRIP1 - dq, keyBH = me + STRIP1 + E24 + (1ull << 24);
        bool okA = (tb >= wlo) & !far, okB = (te <= whi) & !far;
        if (__any(!(okA & okB))) {
            // a neighbour strip that sticks out of the staged range is clipped to it when the staged part provably holds the q window
            const u64k f = lw[wlo], l = lw[whi - 1];
            K2STAT(4, !far && !(okA & okB));
            if (!okA & !far) { okA = (f >= (me & ~((1ull << 52) - 1ull)) - STRIP1) & (f < keyAL); tb = wlo; }
            if (!okB & !far) { okB = (l >= keyBH) & (l < (me & ~((1ull << 52) - 1ull)) + 2 * STRIP1); te = whi; }
        }
        const int longest = max(b + off - tb, te - e);
        int hja = -1, hjb = -1;
        bool deferred = false;
        if (okA & okB) {
            int ja, jb;
            if (!__any(longest > 31)) { ja = k2k_first_ge<5>(lw, tb, keyAL); jb = k2k_first_ge<5>(lw, e, keyBL); }
            else if (!__any(longest > 127)) { ja = k2k_first_ge<7>(lw, tb, keyAL); jb = k2k_first_ge_clamped<7>(lw, e, last, keyBL); }
            else if (!__any(longest > 255)) { ja = k2k_first_ge_clamped<8>(lw, tb, last, keyAL); jb = k2k_first_ge_clamped<8>(lw, e, last, keyBL); }
            else if (!__any(longest > 511)) { ja = k2k_first_ge_clamped<9>(lw, tb, last, keyAL); jb = k2k_first_ge_clamped<9>(lw, e, last, keyBL); }
            else if (!__any(longest > 1023)) { ja = k2k_first_ge_clamped<10>(lw, tb, last, keyAL); jb = k2k_first_ge_clamped<10>(lw, e, last, keyBL); }
            else { ja = k2k_first_ge_clamped<12>(lw, tb, last, keyAL); jb = k2k_first_ge_clamped<12>(lw, e, last, keyBL); }
            const int ka = k2k_first_ge<6>(lw, ja, keyAH), kb = k2k_first_ge<6>(lw, jb, keyBH);
            hja = ja; hjb = jb;
            K2_ABL(512);
            const int na = ka - ja, nb = kb - jb;
            // class counts of both windows and the query's own third (its increment of the running counters)
            const unsigned cja = (unsigned)lw[ja], cka = (unsigned)lw[ka], cjb = (unsigned)lw[jb], ckb = (unsigned)lw[kb];
            const unsigned inc = ((unsigned)lw[li + 1] - (unsigned)lw[li]) & 0xffffffu;
            const unsigned dA = (cka - cja) & 0xffffffu, dB = (ckb - cjb) & 0xffffffu;
            const int A1 = (int)(dA & 0xfffu), A2 = (int)(dA >> 12), B1 = (int)(dB & 0xfffu), B2 = (int)(dB >> 12);
            const bool f1 = (inc & 1u) != 0, f2 = (inc >> 12) != 0;
            // one strip below: neighbours have r_j >= r_i -- the higher thirds surely, the own third perhaps; one strip above: mirrored
            const int sureA = f2 ? 0 : (f1 ? A2 : A1), uncA = f2 ? A2 : (f1 ? A1 - A2 : na - A1);
            const int sureB = f2 ? nb - B2 : (f1 ? nb - B1 : 0), uncB = f2 ? B2 : (f1 ? B1 - B2 : nb - B1);
            const int low = c + sureA + sureB, up = low + uncA + uncB;
            const bool capped = (na >= K2K_HCAP) | (nb >= K2K_HCAP);
            // count in [low, up] (up only if both windows were searched to their ends).  Core at every served minPts: done.  No served
            // minPts in (low, up]: every test reads the same from up as from the count (up <= 1 = isolated: up is a true upper bound).
            if (low >= minPts) c = low;
            else if (!capped && up < (int)l_next[low]) c = up;
            else {
                int na2 = na, nb2 = nb;
                K2STAT(7, capped);
                if (capped) {
                    // a window longer than the shallow search: its true end (clamped to the array: okA / okB say the window ends inside)
                    if (na >= K2K_HCAP) na2 = k2k_first_ge_clamped<12>(lw, ja, last, keyAH) - ja;
                    if (nb >= K2K_HCAP) nb2 = k2k_first_ge_clamped<12>(lw, jb, last, keyBH) - jb;
                }
                // the slots are numbered over the lanes whose window sizes fit the entry: the deferred entries of a round are then the
                // first ones of that numbering (no gaps in the list: phase 3 reads entries 0 .. n3 - 1)
                const bool fits = (na2 < 1024) & (nb2 < 1024);
                const unsigned long long bal = __ballot(fits);
                const int slot = n3 + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                bool room = fits && 2 * slot + 1 < (h - lane) + 64;
#ifdef CLOOPS_DEVEL
                if (g.dbg & 8192) room = false;           // (developer knob: every walk in place)
#endif
                if (room) {
                    // (tix 10 bits | c 7 | na 10 ; ja 12 | jb 12 ... nb rides in the spare bits)
                    my_list[2 * slot] = (unsigned)tix | ((unsigned)c << 10) | ((unsigned)na2 << 17) | ((unsigned)(nb2 & 31) << 27);
                    my_list[2 * slot + 1] = (unsigned)ja | ((unsigned)jb << 12) | ((unsigned)(nb2 >> 5) << 24);
                    deferred = true;
                    K2STAT(5, true);
                } else { K2STAT(6, true); c = walk(c, tix, ja, na2, jb, nb2); }
            }
        } else {
            // a neighbour strip reaches outside the staged window (pile-up) or lies beyond the key clamp: global memory
            K2STAT(3, true);
            const int pi = sa[t0 + tix], qlo = (int)qi - eps, qhi = (int)qi + eps;
            const int gtb = gtb0, ge = e - off, gte = gte0;
            if (gtb < b) {
                const int j = lower_bound_4(sv, gtb, b, qlo);
                c = k2_count_glb<false, 8>(sv, sa, j, b, qhi, pi, peps, minPts, c);
            }
            if (c < minPts && ge < gte) {
                const int j = lower_bound_4(sv, ge, gte, qlo);
                c = k2_count_glb<false, 8>(sv, sa, j, gte, qhi, pi, peps, minPts, c);
            }
        }
        n3 += __popcll(__ballot(deferred));
        if (!deferred) emit(tix, li, c, hja, hjb);
    }
    n3 = __builtin_amdgcn_readfirstlane(dpp_reduce_wave(n3, OpMax()));
    // ---- phase 3: the candidate walks of the deferred PETs, by the whole workgroup as one list --------------------------------
    if (lane == 0) l_st[K2F_NS + wv] = n3;
    __syncthreads();
    const int c0 = l_st[K2F_NS], c1 = c0 + l_st[K2F_NS + 1], c2 = c1 + l_st[K2F_NS + 2];
    int ntot = c2 + l_st[K2F_NS + 3];
#ifdef CLOOPS_DEVEL
    if (g.dbg & 2048) ntot = 0;
#endif
    // FOUR lanes per deferred PET: a walk is a chain of round trips to the L2 (the strip coordinates are not staged), a workgroup has a
    // few tens of deferred PETs for its 256 threads, and what it waits for is the longest chain -- lane k of a quad takes the
    // candidates 4k .. 4k + 3 of every 16 of both windows (8 loads in flight), the quad adds up
    for (int gi = (int)threadIdx.x; gi < 4 * ntot; gi += K2F_TPB) {
        const int ei = gi >> 2, sub = gi & 3;
        const int w = (ei >= c0) + (ei >= c1) + (ei >= c2);
        const int h = ei - (w == 0 ? 0 : (w == 1 ? c0 : (w == 2 ? c1 : c2)));
        const unsigned int* wl = l_list + w * (64 * U);
        const unsigned e0 = wl[2 * h], e1 = wl[2 * h + 1];
        const int tix = (int)(e0 & 0x3ffu), li = HALO + tix;
        const int ja = (int)(e1 & 0xfffu), jb = (int)((e1 >> 12) & 0xfffu);
        const int na = (int)((e0 >> 17) & 0x3ffu), nb = (int)((e0 >> 27) | ((e1 >> 24) << 5));
        const int pi = sa[t0 + tix], plo = pi - peps, phi = pi + peps;
        const int* __restrict__ ga = sa + (t0 - HALO) + ja;
        const int* __restrict__ gb = sa + (t0 - HALO) + jb;
        int add = 0;
#pragma unroll 1
        for (int j = 4 * sub; (j < na) | (j < nb); j += 16) {
            int va[4], vb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { va[k] = ga[j + k]; vb[k] = gb[j + k]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) add += (((j + k < na) & (va[k] >= plo)) ? 1 : 0) + (((j + k < nb) & (vb[k] <= phi)) ? 1 : 0);
        }
        add += CL_DPP(add, 0xb1, 0xf);                  // quad_perm [1,0,3,2]
        add += CL_DPP(add, 0x4e, 0xf);                  // quad_perm [2,3,0,1]
        const int c = min((int)((e0 >> 10) & 0x7fu) + add, minPts);      // (a core PET's word is saturated at minPts)
        if (sub == 0) emit(tix, li, c, ja, jb);
    }
}

// ------------------------------------------------------------------------------------------
// host side: pick the tile shape and launch
// ------------------------------------------------------------------------------------------
int cl_launch_region(hipStream_t stream, const GridParams& g, int n, int run_m, bool exact, const int* sv, const int* sa,
                     const int* strip_start, const int* tile_s0, int* cnt)
{
        const int m1 = g.minPts - 1;
        if (!exact && m1 >= 1 && m1 <= 127) {
            // clustering form: tile / halo picked from the mean strip population (long strips need a wide window)
            const long long avg = (long long)n / std::max(1, g.S);
            int shape = avg <= 40 ? 0 : (avg <= 400 ? 1 : 2);
#ifdef CLOOPS_DEVEL
            if (const char* e = getenv("CLOOPS_K2_SHAPE")) shape = atoi(e);
#endif
            int padlds = 0;                                 // developer knob: dynamic LDS on top of the static arrays (occupancy experiments)
#ifdef CLOOPS_DEVEL
            if (const char* e = getenv("CLOOPS_K2_PADLDS")) padlds = atoi(e);
#endif
#define K2F_LAUNCH(UU, HH)                                                                                              \
            {                                                                                                           \
                const int tile = K2F_TPB * UU, ntiles = nblocks(std::max(1, run_m), tile), run = std::max(1, 2048 / tile); \
                const int grid = ((ntiles + 8 * run - 1) / (8 * run)) * (8 * run);                                      \
                if (g.cut > 0) hipLaunchKernelGGL((k_region_core<UU, HH, true>), dim3(grid), dim3(K2F_TPB), padlds, stream, g, ntiles, \
                                   sv, sa, strip_start, tile_s0, cnt);                            \
                else hipLaunchKernelGGL((k_region_core<UU, HH, false>), dim3(grid), dim3(K2F_TPB), padlds, stream, g, ntiles, \
                                   sv, sa, strip_start, tile_s0, cnt); \
            }
            // the dense shapes run the sorted-key form when q fits its 28-bit field (q + eps + 1 < 2^28: every human chromosome)
            bool keys = shape >= 1 && shape <= 2 && (long long)g.qtop + g.eps + 1 < (1LL << 28);
#ifdef CLOOPS_DEVEL
            if (getenv("CLOOPS_K2_OLD")) keys = false;
#endif
#define K2K_LAUNCH(UU, HH)                                                                                              \
            {                                                                                                           \
                const int tile = K2F_TPB * UU, ntiles = nblocks(std::max(1, run_m), tile), run = std::max(1, 2048 / tile); \
                const int grid = ((ntiles + 8 * run - 1) / (8 * run)) * (8 * run);                                      \
                if (g.cut > 0) hipLaunchKernelGGL((k_region_keys<UU, HH, true>), dim3(grid), dim3(K2F_TPB), padlds, stream, g, ntiles, \
                                   sv, sa, strip_start, tile_s0, cnt);                            \
                else hipLaunchKernelGGL((k_region_keys<UU, HH, false>), dim3(grid), dim3(K2F_TPB), padlds, stream, g, ntiles, \
                                   sv, sa, strip_start, tile_s0, cnt); \
            }
            if (keys) {
                if (shape == 1) K2K_LAUNCH(4, 512) else K2K_LAUNCH(4, 1024)
                return CL_OK;
            }
#undef K2K_LAUNCH
            // window = tile + 2 * halo entries; shapes keep it a multiple of 1024 (every thread stages whole 16-byte slots)
            switch (shape) {
            case 0: K2F_LAUNCH(3, 128) break;
            case 1: K2F_LAUNCH(4, 512) break;
            case 2: K2F_LAUNCH(4, 1024) break;
#ifdef CLOOPS_DEVEL
            case 3: K2F_LAUNCH(2, 256) break;
            case 4: K2F_LAUNCH(6, 256) break;
            case 5: K2F_LAUNCH(2, 768) break;
            case 6: K2F_LAUNCH(1, 384) break;
#endif
            default: K2F_LAUNCH(4, 1024) break;
            }
#undef K2F_LAUNCH
#ifdef CLOOPS_DEVEL
            if (getenv("CLOOPS_K2_CLOCK")) {
                static std::vector<unsigned> h(1 << 21);
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_k2t), h.size() * 4);
                double sum[6] = {0, 0, 0, 0, 0, 0}; long long nw = 0, s1 = 0, s2 = 0;
                for (size_t w = 0; w < (1u << 18); ++w) if (h[w * 8 + 7]) { ++nw; for (int k = 0; k < 6; ++k) sum[k] += h[w * 8 + k]; s1 += h[w * 8 + 6] >> 16; s2 += h[w * 8 + 6] & 0xffff; }
                fprintf(stderr, "[k2 clock] undecided after phase 0: %lld, after the own strip: %lld (of %d rows)\n", s1, s2, n);
                fprintf(stderr, "[k2 clock] %lld waves; mean cycles per wave: scalar %.0f | stage %.0f | barrier1 %.0f | phase0 %.0f | barrier2 %.0f | hard %.0f\n",
                        nw, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw);
                std::fill(h.begin(), h.end(), 0u);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k2t), h.data(), h.size() * 4);
            }
#endif
        } else {
            const int ntiles = nblocks(n, K2_TPB);
            const int grid = ((ntiles + 8 * K2_RUN - 1) / (8 * K2_RUN)) * (8 * K2_RUN);
            if (exact) hipLaunchKernelGGL(k_region_count<true>, dim3(grid), dim3(K2_TPB), 0, stream, g, ntiles, n, sv, sa, strip_start, cnt);
            else hipLaunchKernelGGL(k_region_count<false>, dim3(grid), dim3(K2_TPB), 0, stream, g, ntiles, n, sv, sa, strip_start, cnt);
        }
    return CL_OK;
}

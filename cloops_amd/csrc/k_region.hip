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

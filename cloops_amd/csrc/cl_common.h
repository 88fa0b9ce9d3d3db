// cl_common.h -- shared by the translation units of libcloops_hip.so: types, error plumbing, GridParams, wave / LDS
// helpers.  Device helpers are __forceinline__, so every TU carries its own copy (no relocatable device code).
#pragma once
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <climits>
#include <ctime>
#include <string>
#include <vector>
#include <algorithm>
#include <iterator>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/cloops_hip.h"

#define CL_VERSION_NUM 100   // 0.1.0

typedef unsigned long long u64;
typedef unsigned int u32;

// rocPRIM onesweep radix sort with 9-bit digits: the 45 significant key bits of a chr1-sized
// chromosome take 5 passes instead of 6 (measured on MI355X, 5 M pairs: 344 us vs 405 us default)
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                   rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, 9,
                                                                       rocprim::block_radix_rank_algorithm::match>>
    SortConfig;


// ------------------------------------------------------------------------------------------
// error plumbing (g_err lives in cloops_hip.hip)
// ------------------------------------------------------------------------------------------
extern thread_local std::string g_err;

static inline int fail(int code, const char* what, const char* detail = nullptr)
{
    g_err = what;
    if (detail) { g_err += ": "; g_err += detail; }
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail(CL_ERR_HIP, #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
#define TPB 256

enum { ST_LIVE = 0, ST_DEAD = 1, ST_UNKNOWN = 2 };          // variant-2 release state of a component
enum { CTR_NU = 0, CTR_NREC = 1, CTR_OVERFLOW = 2, CTR_NROOT = 3, CTR_NFLAG = 4, CTR_NOVF = 5 };         // device counters
enum { CTR_TICKET_A = 50, CTR_TICKET_B = 51 };          // "last workgroup" tickets of the in-kernel scans: zero between kernels, never cleared with the counters

struct GridParams {
    int eps;      // cell / strip width (cDBSCAN.py:29, cDBSCAN2.py:30: cw = eps)
    int minPts;
    int cut;      // pipe.py:59-63 pre-filter, 0 = off
    int A0;       // offset subtracted from the STRIP coordinate  (0 for variant 2: absolute cells)
    int V0;       // offset subtracted from the IN-STRIP (sorted) coordinate (0 for variant 2)
    int swap;     // 0: strips are bands of a = Y-X ordered by v = X+Y;  1: bands of v ordered by a
    int s0;       // strip index of the first table row
    int S;        // number of strips in the table; key strip S marks filtered rows
    int variant;
    int dbg;      // developer knobs (CLOOPS_DBG env), 0 in production
    int dbg2;     // more of them (CLOOPS_DBG2)
    u32 magic;    // strip(a) = a / eps by multiply-shift (Granlund-Montgomery, exact for all u32)
    int sh1, sh2;
    int qbits;    // sort key = strip << (qbits+rbits) | q << rbits | (p mod eps): both coordinates ride
    int rbits;    //   in the key, so the sorted (q,p) arrays are DECODED, not gathered through row ids;
                  //   the radix sort skips the low rbits (they are payload, not order)
    u32 tmask[4]; // K2 (clustering form): the minPts values the words of this run have to serve, bit t - 1 = minPts t (2 .. 128).  A
                  //   one-off run: its own minPts alone.  A run whose words later runs of the eps re-use (count cache of the handle):
                  //   every minPts that will follow (cl_set_count_thresholds), or all of [floor, minPts] (cl_set_count_floor).  The
                  //   count a non-core PET's word holds is only as exact as those tests need: for every t of the set,
                  //   stored >= t  <=>  count >= t (an upper bound of the count otherwise; <= 1 still means "nothing within eps")
    int tgap;     //   the widest gap of that set: max over c in [1, minPts) of (smallest served minPts above c) - c
    int qmin;     // K2 (clustering form): PETs with q below it get no word (level 4: the words of an eps are made on the base layout by a run
                  //   under a cut -- the PETs its cut removes are in the cut band of every run that could ever read their words)
    int qtop;     // the largest in-strip coordinate q of the chromosome (k_region_keys packs q into 28 bits of a sorted 64-bit key)
    int peps;     // 1 << rbits.  The kernels never see p itself but its ORDER-PRESERVING re-encoding
                  //   sp = strip << rbits | (p mod eps)   (the strip and remainder fields of the sort key):
                  //   strip(p) = sp >> rbits (no division), and |p_j - p_i| <= eps  <=>  |sp_j - sp_i| <= peps
                  //   (same strip: always; adjacent strips: both compare the remainders; two or more strips apart: never).
};

__device__ __forceinline__ int sat_add(int a, int b)
{
    long long s = (long long)a + (long long)b;
    return s > INT_MAX ? INT_MAX : (s < INT_MIN ? INT_MIN : (int)s);
}

// ---- wave64 reductions on the DPP network (no LDS crossbar traffic, a handful of VALU instructions) ------------
// quad_perm [1,0,3,2], [2,3,0,1], row_shr:4, row_shr:8 leave every 16-lane row's total in its lane 15; row_bcast:15 then
// lanes 31 / 63 hold the totals of lanes 0..31 / 32..63; row_bcast:31 completes lane 63.  min / max are idempotent, so
// lanes that receive nothing combine with their own value.
#define CL_DPP(v, ctrl, rowmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rowmask), 0xf, false)
template <typename Op>
__device__ __forceinline__ int dpp_reduce_halves(int v, Op op)       // result: lane 31 <- lanes 0..31, lane 63 <- lanes 32..63
{
    v = op(v, CL_DPP(v, 0xb1, 0xf));
    v = op(v, CL_DPP(v, 0x4e, 0xf));
    v = op(v, CL_DPP(v, 0x114, 0xf));
    v = op(v, CL_DPP(v, 0x118, 0xf));
    v = op(v, CL_DPP(v, 0x142, 0xa));
    return v;
}
template <typename Op>
__device__ __forceinline__ int dpp_reduce_wave(int v, Op op)         // wave-uniform result
{
    v = dpp_reduce_halves(v, op);
    v = op(v, CL_DPP(v, 0x143, 0xc));
    return __builtin_amdgcn_readlane(v, 63);
}
struct OpMin { __device__ __forceinline__ int operator()(int a, int b) const { return min(a, b); } };
struct OpMax { __device__ __forceinline__ int operator()(int a, int b) const { return max(a, b); } };

// first index in [lo,hi) with sv[idx] >= val
__device__ __forceinline__ int lower_bound_i(const int* __restrict__ sv, int lo, int hi, int val)
{
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (sv[mid] < val) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index in [lo,hi) with sv[idx] > val
__device__ __forceinline__ int upper_bound_i(const int* __restrict__ sv, int lo, int hi, int val)
{
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (sv[mid] <= val) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int div_eps(const GridParams& g, int arel)
{
    const u32 n = (u32)arel;                       // arel >= 0 by construction
    const u32 t1 = __umulhi(g.magic, n);
    return (int)((t1 + ((n - t1) >> g.sh1)) >> g.sh2);
}
__device__ __forceinline__ int strip_of(const GridParams& g, int sp) { return sp >> g.rbits; }      // sp: see GridParams

// ---- lock-free union-find with randomised linking ------------------------------------------
// Every node has a fixed pseudo-random priority (a bijective hash of its index); a root is only
// ever hooked under a root of HIGHER priority, so the forest is acyclic whatever the interleaving
// and its expected depth is logarithmic even for a component that is a 50 000-strip long path
// (the self-ligation diagonal at large eps: linking by smaller index made that a 50 000-deep list
// whose first traversal alone cost 8 ms).  Which member ends up as the root is irrelevant -- ids,
// keys and sizes are all reduced over the members.
// parent[] is read with PLAIN (L1-cacheable) loads: a stale value is always an earlier parent of
// the same node, i.e. still an ancestor, so a find that stops early merely returns a non-root
// ancestor.  Only the hook is an atomic: atomicCAS succeeds only on a true root, and when it fails it
// returns the true parent, whose priority is strictly higher -- every retry makes progress.
// (Agent-scope atomic loads here serialise millions of lanes on the one L2 channel holding a giant
// component's root: 77 ms vs 1 ms on a 16 M-PET chromosome.)
__device__ __forceinline__ unsigned uf_prio(int x)
{
    unsigned h = (unsigned)x;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;      // bijective
    return h;
}
__device__ __forceinline__ int uf_find(int* parent, int x)
{
    for (;;) {
        int p = parent[x];
        if (p == x) return x;
        int gp = parent[p];
        if (gp == p) return p;
        parent[x] = gp;                             // path halving (benign race: gp is an ancestor)
        x = gp;
    }
}
__device__ __forceinline__ void uf_unite(int* parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (uf_prio(a) > uf_prio(b)) { int t = a; a = b; b = t; }      // a = lower priority: it goes under b
        int old = atomicCAS(parent + a, a, b);
        if (old == a) return;
        a = old;                                    // not a root any more: continue from its true parent
    }
}
__device__ __forceinline__ int uf_find_ro(const int* __restrict__ parent, int x)
{   // after the union kernel has completed (kernel boundary = coherent), plain loads
    int p = parent[x];
    while (p != x) { x = p; p = parent[x]; }
    return x;
}

// ------------------------------------------------------------------------------------------
// K2: region query  (cDBSCAN.py:186-205 regionQuery / cDBSCAN2.py:304-334 neighbour count)
// ------------------------------------------------------------------------------------------
// 4-way lower bound: three independent probes per step -- half the dependent memory round
// trips of a bisect (the global path is latency bound: one L2/HBM round trip per step).
__device__ __forceinline__ int lower_bound_4(const int* pv, int lo, int hi, int val)
{
    while (hi - lo > 4) {
        const int q = (hi - lo) >> 2;
        const int m1 = lo + q, m2 = m1 + q, m3 = m2 + q;
        const int v1 = pv[m1], v2 = pv[m2], v3 = pv[m3];
        if (v1 >= val) hi = m1;
        else if (v2 >= val) { lo = m1 + 1; hi = m2; }
        else if (v3 >= val) { lo = m2 + 1; hi = m3; }
        else lo = m3 + 1;
    }
    while (lo < hi && pv[lo] < val) ++lo;
    return lo;
}

// inclusive scan over the 256 threads of a workgroup (all of them call); red: 4 ints of LDS; total = the workgroup's sum
__device__ __forceinline__ int wg256_inclusive_scan(int v, int* red, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(v, d); v += lane >= d ? t : 0; }
    if (lane == 63) red[wave] = v;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int x = red[w]; off += w < wave ? x : 0; tot += x; }
    __syncthreads();
    total = tot;
    return v + off;
}
// The tail of a two-level exclusive scan inside the kernel that made its input (no launch of its own, nothing spins): every
// workgroup has written the sum of its 256 elements to bsum[blockIdx.x] and takes a ticket; the LAST one to arrive scans the
// block sums -> boff[0 .. nblk] (boff[nblk] = the total) and puts the ticket back to zero.  Element e of the scan is then
// loc[e] (exclusive inside its workgroup) + boff[e >> 8].  Called by all 256 threads of every workgroup.
__device__ __forceinline__ void scan_tail_last_block(int mysum, int* __restrict__ bsum, int* __restrict__ boff, int* ticket, int* red)
{
    __shared__ int l_is_last;
    if (threadIdx.x == 0) {
        // The block sum goes out as a device-scope ATOMIC (performed at the point all XCDs agree on) whose return is awaited
        // before the ticket is taken; the last workgroup reads the sums with atomic loads.  A release fence instead would
        // write back the whole L2 of the XCD (each of the eight has its own): 10+ us per kernel.
        const int old = __hip_atomic_exchange(&bsum[blockIdx.x], mysum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" :: "v"(old) : "memory");
        l_is_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!l_is_last) return;
    const int nblk = (int)gridDim.x;
    int carry = 0;
    for (int base = 0; base < nblk; base += 256) {
        const int k = base + (int)threadIdx.x;
        const int v = k < nblk ? __hip_atomic_load(&bsum[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int tot;
        const int incl = wg256_inclusive_scan(v, red, tot);
        if (k < nblk) boff[k] = carry + incl - v;
        carry += tot;
    }
    if (threadIdx.x == 0) { boff[nblk] = carry; __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

struct LdsPairs {       // LDS window of (q = in-strip coord, p = strip coord), addressed by GLOBAL sorted index
    const int2* a; int base;
    __device__ __forceinline__ int2 operator[](int j) const { return a[j - base]; }
    __device__ __forceinline__ int qat(int j) const { return a[j - base].x; }
};
struct LdsSoA {         // same window as two int arrays: the searches only read q, and consecutive
    const int* q; const int* p; int base;          // dwords spread over all LDS banks (pairs: every other bank)
    __device__ __forceinline__ int2 operator[](int j) const { return make_int2(q[j - base], p[j - base]); }
    __device__ __forceinline__ int qat(int j) const { return q[j - base]; }
};
struct LdsInts {
    const int* a; int base;
    __device__ __forceinline__ int operator[](int j) const { return a[j - base]; }
};

// Branch-free bounded searches on an LDS window of (q,p) pairs: fixed 8 steps, no divergence
// (a wave pays the LONGEST trip count of its lanes, so data-dependent loops cost far more
// instructions than the average lane needs).  Valid for hi - lo <= 255.
// first idx in [lo,hi) with w[idx].x >= val (or hi)
template <int STEPS = 8, typename W>
__device__ __forceinline__ int lds_lower_bound8(const W& w, int lo, int hi, int val)
{
    int pos = lo;
#pragma unroll
    for (int step = 1 << (STEPS - 1); step >= 1; step >>= 1) {
        const int idx = pos + step - 1;
        const int v = w.qat(min(idx, hi - 1));
        pos = (idx < hi && v < val) ? pos + step : pos;
    }
    return pos;
}
// first idx in [lo,hi) with w[idx].x > val (or hi)
template <int STEPS = 8, typename W>
__device__ __forceinline__ int lds_upper_bound8(const W& w, int lo, int hi, int val)
{
    int pos = lo;
#pragma unroll
    for (int step = 1 << (STEPS - 1); step >= 1; step >>= 1) {
        const int idx = pos + step - 1;
        const int v = w.qat(min(idx, hi - 1));
        pos = (idx < hi && v <= val) ? pos + step : pos;
    }
    return pos;
}

// Count the candidates j of [j,te) with q[j] <= qhi (q ascending) and |p[j]-pi| <= eps, in chunks
// whose loads are all issued before the first compare (an element-at-a-time `while (q <= qhi)`
// loop costs one full memory latency per candidate).
template <bool EXACT, int CH, typename W>
__device__ __forceinline__ int k2_count_lds(const W& w, int j, int te, int qhi, int pi, int eps, int minPts, int c)
{
    while (j < te) {
        int2 v[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) v[k] = w[min(j + k, te - 1)];
        bool out = false;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const bool in = (j + k < te) && (v[k].x <= qhi);
            out |= !in;
            const int da = v[k].y - pi;
            c += (in && (da < 0 ? -da : da) <= eps) ? 1 : 0;
        }
        if (out || (!EXACT && c >= minPts)) break;
        j += CH;
    }
    return c;
}
template <bool EXACT, int CH>
__device__ __forceinline__ int k2_count_glb(const int* __restrict__ pq, const int* __restrict__ pp, int j, int te,
                                            int qhi, int pi, int eps, int minPts, int c)
{
    while (j < te) {
        int v[CH], a[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) { const int idx = min(j + k, te - 1); v[k] = pq[idx]; a[k] = pp[idx]; }
        bool out = false;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const bool in = (j + k < te) && (v[k] <= qhi);
            out |= !in;
            const int da = a[k] - pi;
            c += (in && (da < 0 ? -da : da) <= eps) ? 1 : 0;
        }
        if (out || (!EXACT && c >= minPts)) break;
        j += CH;
    }
    return c;
}

// Workgroup compaction: slot list of the threads with `active`; returns their number.  Whole
// waves fall out of the expensive phase instead of running it at partial lane occupancy.
template <int NT = TPB>
__device__ __forceinline__ int block_compact(bool active, short* l_list, int* l_wcount)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(active);
    if (lane == 0) l_wcount[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { const int c = l_wcount[k]; off += (k < wv) ? c : 0; total += c; }
    if (active) l_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)threadIdx.x;
    __syncthreads();
    return total;
}

// same, with `between()` executed by every thread between the two barriers (stores that may complete late)
template <int NT, typename F>
__device__ __forceinline__ int block_compact_with(bool active, short* l_list, int* l_wcount, F&& between)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(active);
    if (lane == 0) l_wcount[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) { const int c = l_wcount[k]; off += (k < wv) ? c : 0; total += c; }
    if (active) l_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (short)threadIdx.x;
    between();
    __syncthreads();
    return total;
}

#define UNION_CH 8       // candidates fetched per round trip in the long-strip scan of k_union_cores
#define K2_TPB 256
#define K2_HALO 128
#define K2_WIN (K2_TPB + 2 * K2_HALO)
#define K2_SPAN 120      // own-strip window searched branch-free within +-K2_SPAN positions
#define K2_RUN 8         // consecutive tiles given to one XCD (halo reuse in that XCD's L2)

// ---- K2 constants the other TUs see (tile / pad sizes, the hint word k_region_core leaves for k_border) ----
// The sorted arrays sv / sa carry SORT_PAD sentinel entries in front of index 0 and behind index n-1 (left: q = 0,
// sp = INT_MIN; right: q = sp = INT_MAX -- "in no strip"), written once when the workspace is allocated: a tile
// window is staged with unpredicated 16-byte loads, no bounds logic at all.
#define K2F_TPB 256
#define K2F_NS 256        // staged strip-table slice: strips s0-1 .. s0+254 of the tile's first strip s0
#define K2F_SLACK 128     // LDS entries behind the window that unclamped search probes may touch
#define SORT_PAD 4224     // >= largest tile + largest halo + slack
// The word K2 leaves per PET (cnt[]; every consumer decodes it with cw_count / cw_core):
//   w >= 0   a neighbour count, saturated: k_region_core stores minPts (or more) for a core PET, k_region_count the count;
//   w <  0   a PET that is NOT core at the minPts the word was made with: bits 24..30 its neighbour count (itself included;
//            exact from GridParams::floor up, an upper bound below), bits 0..11 / 12..23 how many sorted positions back its
//            q window in strip s-1 starts / how many positions ahead the one in strip s+1 starts (all ones = no hints).
// A word stays valid for every minPts in [floor, minPts of the run that made it]: the count cache of a handle re-uses the
// words of the first run of an eps for the runs that follow (cLoops/pipe.py:247-250 walks minPts inside eps).
#define K2H_BITS 12
#define K2H_MASK 0xfffu
#define K2H_NONE 0x00ffffffu
#define K2W_CSHIFT 24
__device__ __forceinline__ int cw_count(int w) { return w >= 0 ? w : (int)(((unsigned)w >> K2W_CSHIFT) & 0x7fu); }
__device__ __forceinline__ bool cw_core(int w, int minPts) { return cw_count(w) >= minPts; }

// Where the K2 word of sorted position i of the running layout lives.  A run that made its own words (or runs on the very
// layout they were made on): rc[i] (D == nullptr).  A run that re-uses the words of an earlier run of this eps under another
// cut (count cache, cl_chrom::rc): the PETs of the cut band (q < bandq) have a fresh word in band[i] (k_region_band); every
// other PET has its word at its place in the layout the words were made on, rc[i + D[strip]] (D[s] = how many more PETs
// this run's cut removes up to and including strip s than that run's cut did), with the two hint fields counted in THAT
// layout: they shift by dpre[s] / dpre[s + 1] (what this cut removes from the PET's own strip / the strip above beyond
// what that cut removed); hints that leave their fields are dropped (K2H_NONE: k_border then searches for itself).
struct WordSrc {
    const int* rc; const int* band; const int* D; const int* dpre; int bandq; int rbits;
    __device__ __forceinline__ int raw(int i, int q, int sp) const          // count field valid, hints not shifted
    {
        if (!D) return rc[i];
        return q < bandq ? band[i] : rc[i + D[sp >> rbits]];
    }
    // the same in two steps, so that a thread can have the loads of several PETs in flight: where() -> the table entries, then
    // load(), then shifted()
    struct Where { const int* src; int idx, dA, dB; };
    __device__ __forceinline__ Where where(int i, int q, int sp) const
    {
        Where w; w.src = rc; w.idx = i; w.dA = 0; w.dB = 0;
        if (D) {
            if (q < bandq) w.src = band;
            else { const int s = sp >> rbits; w.idx = i + D[s]; w.dA = dpre[s]; w.dB = dpre[s + 1]; }
        }
        return w;
    }
    __device__ __forceinline__ int shifted(int w, const Where& at) const
    {
        if ((at.dA | at.dB) != 0 && w < 0 && ((unsigned)w & K2H_NONE) != K2H_NONE) {
            const int da = (int)((unsigned)w & K2H_MASK) - at.dA, db = (int)(((unsigned)w >> K2H_BITS) & K2H_MASK) - at.dB;
            const bool ok = (da >= 0) & (da < (int)K2H_MASK) & (db >= 0) & (db < (int)K2H_MASK);
            w = (int)(((unsigned)w & ~K2H_NONE) | (ok ? ((unsigned)da | ((unsigned)db << K2H_BITS)) : K2H_NONE));
        }
        return w;
    }
    __device__ __forceinline__ int word(int i, int q, int sp) const
    {
        if (!D) return rc[i];
        if (q < bandq) return band[i];
        const int s = sp >> rbits;
        int w = rc[i + D[s]];
        if (w < 0 && ((unsigned)w & K2H_NONE) != K2H_NONE) {
            const int da = (int)((unsigned)w & K2H_MASK) - dpre[s], db = (int)(((unsigned)w >> K2H_BITS) & K2H_MASK) - dpre[s + 1];
            const bool ok = (da >= 0) & (da < (int)K2H_MASK) & (db >= 0) & (db < (int)K2H_MASK);
            w = (int)(((unsigned)w & ~K2H_NONE) | (ok ? ((unsigned)da | ((unsigned)db << K2H_BITS)) : K2H_NONE));
        }
        return w;
    }
};

static inline int nblocks(long long n, int tpb = TPB) { return (int)((n + tpb - 1) / tpb); }

// k_region.hip: K2 (neighbour counts / core decision) on the sorted arrays of a run
int cl_launch_region(hipStream_t stream, const GridParams& g, int n, int run_m, bool exact, const int* sv, const int* sa,
                     const int* strip_start, const int* tile_s0, int* cnt);


// cl_band.h -- K2 on the cut band (count cache of the handle, cl_chrom::rc), as device code: the kernel k_band of traversal level 4
// (words and hints in base positions, BASEOUT) and, at levels <= 3, a part of k_cut_copy.
//
// A run that re-uses the K2 words of an earlier run of its eps has to redo the region query (cDBSCAN.py:186-205,
// cDBSCAN2.py:333-334) only for the PETs whose neighbourhood the two cuts treat differently: q < bandq = the larger cut +
// eps.  The sorted order inside a strip is q, so those are the FIRST blen[s].x kept PETs of every strip, and everything
// within eps of them lies among the first blen[s].y (q < bandq + eps) of the strips s-1, s, s+1 -- a few per cent of the
// chromosome, a handful of PETs per strip.  That little work is pure latency (table row -> strip prefixes -> a few LDS
// searches -> one store per PET).  At levels <= 3 it does not get a launch of its own: the first workgroups of the run's compaction
// kernel do it while the others stream the copy (level 4 copies nothing: k_band, 17-35 us), reading the prefixes straight from the BASE layout (the kept PETs of strip
// s start at src0[s] there) and writing the words at the PETs' places in the new one (new strip start + position).
//
// One WAVE takes KB_SB consecutive strips (their table rows live in its lanes, the offsets come from shuffles -- no
// barrier anywhere), stages the KB_SB + 2 strip prefixes as (q, sp) pairs in its own 4 KB of LDS with all loads of a lane in
// flight together, deals the band PETs to its lanes, and every lane runs its six window searches (own strip and strips
// s-1 / s+1: both ends of the q window) branch-free in lockstep.  "Own + everything in both q windows < minPts" settles a
// PET without reading a candidate (the word of a band PET is not kept: an upper bound of its count will do, as in
// k_region_core); the others test the strip coordinate of their candidates until minPts is reached.  Prefixes beyond the
// staging area (pile-ups) are read from global memory.  Same word format as k_region_core (cl_common.h "K2W").
#pragma once
#include "cl_common.h"

#ifndef KB_SB              // (strips per wave; 2 / 4 / 6 / 8 measured 35 / 23 / 20 / 19 us against 17 for 12 on a chr1 run: the cost is per wave, not per strip)
#define KB_SB 12
#endif
#define KB_CAP 512

// the band PETs of one wave's strips, 64 per round.  LDSP: the strip prefixes are staged in lw (else: global memory, base layout)
// BASEOUT: the words go to the PETs' BASE positions with their hints counted in base positions (traversal level 4: the count
// cache lives in base-position space, k_lists.hip); else to their places in the run's layout (new strip start + offset)
template <bool LDSP, bool BASEOUT = false>
__device__ __forceinline__ void band_rounds(int eps, int peps, int minPts, const int2* lw, const int* __restrict__ bq, const int* __restrict__ bsp,
                                            int* __restrict__ cnt, int lane, int nband, int maxlen,
                                            int ssrc /* base index of the strip's first kept PET */, int sg0 /* its place in the new layout */,
                                            int slen, int soff, int sboff, int dbg)
{
    int nsteps = 0;                                     // bisection depth: wave-uniform, from the longest staged prefix
    while ((1 << nsteps) <= maxlen) ++nsteps;
    for (int t0 = 0; t0 < nband; t0 += 64) {
        const int t = t0 + lane;
        const bool act = t < nband;
        int k = 1;
#pragma unroll
        for (int step = 8; step >= 1; step >>= 1) {         // (largest k in [1, KB_SB] with sboff[k] <= t; KB_SB <= 16)
            const int cand = k + step;
            const int v = __shfl(sboff, min(cand, KB_SB));
            k = (cand <= KB_SB && v <= t) ? cand : k;
        }
        // (every shuffle is executed by the whole wave: a lane that has no PET still serves its table row)
        const int idx = t - __shfl(sboff, k);
        const int g0 = __shfl(sg0, k), len = __shfl(slen, k);
        const int ga = __shfl(sg0, k - 1), lena = __shfl(slen, k - 1);
        const int gb = __shfl(sg0, k + 1), lenb = __shfl(slen, k + 1);
        const int b0 = LDSP ? __shfl(soff, k) : __shfl(ssrc, k);          // segment bases in the space that is read
        const int ba = LDSP ? __shfl(soff, k - 1) : __shfl(ssrc, k - 1);
        const int bb = LDSP ? __shfl(soff, k + 1) : __shfl(ssrc, k + 1);
        const int o0 = BASEOUT ? __shfl(ssrc, k) : g0, oa = BASEOUT ? __shfl(ssrc, k - 1) : ga, ob = BASEOUT ? __shfl(ssrc, k + 1) : gb;      // ... in the space that is written
        const int na = act ? lena : 0, nb = act ? lenb : 0, n0 = act ? len : 0;
        const int ig = o0 + idx;
        auto qat = [&](int pos) { return LDSP ? lw[pos].x : bq[pos]; };
        auto pat = [&](int pos) { return LDSP ? lw[pos].y : bsp[pos]; };
        const int qi = act ? qat(b0 + idx) : 0, pi = act ? pat(b0 + idx) : 0;
        const int qlo = qi - eps, qhi = qi + eps, plo = pi - peps, phi = pi + peps;
        // six branch-free bisections in lockstep over [0, n): position of the first entry with q >= qlo (lo*) / q > qhi (hi*)
        int lo0 = 0, hi0 = 0, loa = 0, hia = 0, lob = 0, hib = 0;
        for (int step = 1 << (nsteps > 0 ? nsteps - 1 : 0); step >= 1; step >>= 1) {
            const int p0 = lo0 + step - 1, p1 = hi0 + step - 1, p2 = loa + step - 1, p3 = hia + step - 1, p4 = lob + step - 1, p5 = hib + step - 1;
            const int v0 = qat(b0 + min(p0, max(n0 - 1, 0))), v1 = qat(b0 + min(p1, max(n0 - 1, 0)));
            const int v2 = qat(ba + min(p2, max(na - 1, 0))), v3 = qat(ba + min(p3, max(na - 1, 0)));
            const int v4 = qat(bb + min(p4, max(nb - 1, 0))), v5 = qat(bb + min(p5, max(nb - 1, 0)));
            lo0 = (p0 < n0 && v0 < qlo) ? lo0 + step : lo0;
            hi0 = (p1 < n0 && v1 <= qhi) ? hi0 + step : hi0;
            loa = (p2 < na && v2 < qlo) ? loa + step : loa;
            hia = (p3 < na && v3 <= qhi) ? hia + step : hia;
            lob = (p4 < nb && v4 < qlo) ? lob + step : lob;
            hib = (p5 < nb && v5 <= qhi) ? hib + step : hib;
        }
        int c = act ? hi0 - lo0 : minPts;
        const int ja = loa, jb = lob, ka = hia, kb = hib;
        const int ub = c + (ka - ja) + (kb - jb);
        int ia = ja, ib = jb;
        bool ma = act && ub >= minPts && c < minPts && ia < ka, mb = act && ub >= minPts && c < minPts && ib < kb;
        if (act && ub < minPts) c = ub;
        while (__any(ma | mb)) {
            int pa[4], pb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                pa[u] = pat(ba + min(ia + u, max(ka - 1, 0)));
                pb[u] = pat(bb + min(ib + u, max(kb - 1, 0)));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c += (ma && ia + u < ka && pa[u] >= plo) ? 1 : 0;        // one strip below: sp can only be too low
                c += (mb && ib + u < kb && pb[u] <= phi) ? 1 : 0;        // one strip above: only too high
            }
            ia += 4; ib += 4;
            ma = ma && c < minPts && ia < ka; mb = mb && c < minPts && ib < kb;
        }
        if (act) {
            int outv = minPts;
            if (c < minPts) {
                const int da = ig - (oa + ja), db = (ob + jb) - ig;
                const bool ok = (da >= 0) & (da < (int)K2H_MASK) & (db >= 0) & (db < (int)K2H_MASK);
                outv = (int)(0x80000000u | ((unsigned)c << K2W_CSHIFT) | (ok ? ((unsigned)da | ((unsigned)db << K2H_BITS)) : K2H_NONE));
            }
            cnt[ig] = outv;
        }
    }
}

// one wave: the strips [group * KB_SB, group * KB_SB + KB_SB).  lw: KB_CAP pairs of LDS owned by this wave.
template <bool BASEOUT = false>
__device__ __forceinline__ void band_wave(int group, int lane, int2* lw, int S, int eps, int peps, int minPts,
                                          const int* __restrict__ bq, const int* __restrict__ bsp, const int* __restrict__ src0,
                                          const int* __restrict__ sloc, const int* __restrict__ sboffs /* the new strip table in its two-level form: k_cut_strips */,
                                          const int2* __restrict__ blen, int* __restrict__ cnt, int dbg)
{
    const int s_first = group * KB_SB;                  // lane k holds segment k = strip s_first - 1 + k, k = 0 .. KB_SB + 1
    int ssrc = 0, sg0 = 0, slen = 0, snb = 0;
    if (lane < KB_SB + 2) {
        const int s = s_first - 1 + lane;
        const bool ok = s >= 0 && s < S;
        const int sc = min(max(s, 0), S);
        sg0 = sloc[sc] + sboffs[sc >> 8];                // (a strip that does not exist: an empty prefix at its place)
        ssrc = ok ? src0[s] : 0;
        const int2 bl = ok ? blen[s] : make_int2(0, 0);
        slen = bl.y;
        snb = (lane >= 1 && lane <= KB_SB) ? bl.x : 0;
    }
    int incl = snb, inclen = slen, maxlen = slen;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {                  // (KB_SB + 2 <= 16 lanes hold a row)
        const int t = __shfl_up(incl, d), u = __shfl_up(inclen, d), m = __shfl_up(maxlen, d);
        incl += lane >= d ? t : 0; inclen += lane >= d ? u : 0; maxlen = lane >= d ? max(maxlen, m) : maxlen;
    }
    const int sboff = incl - snb;                       // band PETs in front of segment `lane`
    const int soff = inclen - slen;                     // staged pairs in front of segment `lane`
    const int nband = __shfl(incl, KB_SB), total = __shfl(inclen, KB_SB + 1);
    maxlen = __shfl(maxlen, KB_SB + 1);
    if (nband == 0) return;
    if (total > KB_CAP) {                               // pile-ups: everything from global memory
        band_rounds<false, BASEOUT>(eps, peps, minPts, lw, bq, bsp, cnt, lane, nband, maxlen, ssrc, sg0, slen, soff, sboff, dbg);
        return;
    }
    {
        // staged position -> (segment, offset) by a search over the lanes' offsets; the loads of a lane all in flight together
        int2 v[KB_CAP / 64];
        int dst[KB_CAP / 64];
#pragma unroll
        for (int u = 0; u < KB_CAP / 64; ++u) {
            dst[u] = -1;
            if (64 * u >= total) continue;              // (wave-uniform: the shuffles below are executed by the whole wave or not at all)
            const int idx = lane + 64 * u;
            int k = 0;
#pragma unroll
            for (int step = 8; step >= 1; step >>= 1) {
                const int cand = k + step;
                const int o = __shfl(soff, min(cand, KB_SB + 1));
                k = (cand <= KB_SB + 1 && o <= idx) ? cand : k;
            }
            const int gk = __shfl(ssrc, k), ok = __shfl(soff, k);
            const bool in = idx < total;
            const int gi = gk + (idx - ok);
            dst[u] = in ? idx : -1;
            v[u] = in ? make_int2(bq[gi], bsp[gi]) : make_int2(0, 0);
        }
#pragma unroll
        for (int u = 0; u < KB_CAP / 64; ++u) if (dst[u] >= 0) lw[dst[u]] = v[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own LDS writes are done
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    band_rounds<true, BASEOUT>(eps, peps, minPts, lw, bq, bsp, cnt, lane, nband, maxlen, ssrc, sg0, slen, soff, sboff, dbg);
}

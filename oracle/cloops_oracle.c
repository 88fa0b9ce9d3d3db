/*
 * cloops_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A sequential C restatement of the three clustering classes of YaqiangCao/cLoops
 * (reference checkout: /root/reference, v0.93):
 *
 *     cl_oracle_v1     <-  cLoops/cDBSCAN.py      class cDBSCAN      (:6-205)
 *     cl_oracle_v2     <-  cLoops/cDBSCAN2.py     class cDBSCAN      (:7-383)   (production, pipe.py:42)
 *     cl_oracle_block  <-  cLoops/blockDBSCAN.py  class blockDBSCAN  (:6-239)
 *
 * Each function follows the reference's *sequential, visit-order dependent* algorithm
 * (same grid, same visiting order, same queue discipline, same overwrite rules) -- on
 * purpose NOT the order-free closed forms the HIP kernels use, so that GPU-vs-oracle
 * parity is a test of those closed forms and not a tautology.
 *
 * Semantics pinned: Python-3 behaviour of the reference (dicts iterate in insertion
 * order, `/` is true division), the only behaviour observable in the build container
 * (SURVEY.md finding 3).  Pinned against the real reference classes by
 * tests/test_oracle_vs_reference.py (runs wherever /root/reference exists) and by the
 * golden vectors under tests/golden/ (generated from the real classes by
 * tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Interface: coordinates are int64 (row i of the reference's `mat` is
 * [id_i, X[i], Y[i]]; ids are not needed because labels are returned aligned to rows),
 * labels[i] = cluster id or -1 for "absent from the reference's .labels dict".
 * Return value: 0 ok, -1 reference would raise (empty input for v1/block), -2 alloc.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------
 * Insertion-ordered cell dictionary  (the reference's `Gs` / `Grid` dicts: key = (nx,ny),
 * iteration order = first insertion, which Python 3.7+ guarantees).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int64_t *knx, *kny;   /* key of cell c (c = insertion rank) */
    int64_t ncell, cap;
    int64_t *slot;        /* open addressing: slot -> cell or -1 */
    int64_t nslot;        /* power of two */
} celldict;

static uint64_t mix64(uint64_t a, uint64_t b)
{
    uint64_t h = a * 0x9E3779B97F4A7C15ull ^ (b + 0x7F4A7C15D1B54A32ull + (a << 6) + (a >> 2));
    h ^= h >> 31; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
    return h;
}

static int cd_init(celldict *d, int64_t expect)
{
    int64_t ns = 16;
    while (ns < expect * 2 + 2) ns <<= 1;
    d->nslot = ns; d->ncell = 0; d->cap = expect > 16 ? expect : 16;
    d->slot = (int64_t *)malloc(sizeof(int64_t) * ns);
    d->knx = (int64_t *)malloc(sizeof(int64_t) * d->cap);
    d->kny = (int64_t *)malloc(sizeof(int64_t) * d->cap);
    if (!d->slot || !d->knx || !d->kny) return -2;
    for (int64_t i = 0; i < ns; i++) d->slot[i] = -1;
    return 0;
}
static void cd_free(celldict *d) { free(d->slot); free(d->knx); free(d->kny); }

static int64_t cd_find(const celldict *d, int64_t nx, int64_t ny)
{
    uint64_t m = (uint64_t)d->nslot - 1, h = mix64((uint64_t)nx, (uint64_t)ny) & m;
    for (;;) {
        int64_t c = d->slot[h];
        if (c < 0) return -1;
        if (d->knx[c] == nx && d->kny[c] == ny) return c;
        h = (h + 1) & m;
    }
}
/* dict.setdefault(key, ...) : returns the cell index, appending a new cell if unseen.
 * capacity is sized for n points up front, so no rehash is ever needed. */
static int64_t cd_get_or_add(celldict *d, int64_t nx, int64_t ny)
{
    uint64_t m = (uint64_t)d->nslot - 1, h = mix64((uint64_t)nx, (uint64_t)ny) & m;
    for (;;) {
        int64_t c = d->slot[h];
        if (c < 0) break;
        if (d->knx[c] == nx && d->kny[c] == ny) return c;
        h = (h + 1) & m;
    }
    int64_t c = d->ncell++;
    d->knx[c] = nx; d->kny[c] = ny; d->slot[h] = c;
    return c;
}

/* `int(a / cw)` with Python-3 true division: float64 quotient, truncated toward zero
 * (cDBSCAN.py:84-85, cDBSCAN2.py:69-70, blockDBSCAN.py:81-82). */
static int64_t py_int_truediv(int64_t a, int64_t cw)
{
    return (int64_t)((double)a / (double)cw);
}

/* the 8 neighbour offsets in the order of getNearbyGrids / getNearbyCells
 * (cDBSCAN.py:59-60, cDBSCAN2.py:42-43, blockDBSCAN.py:56-57) */
static const int NB8[8][2] = { {0,-1},{0,1},{-1,0},{1,0},{-1,-1},{-1,1},{1,-1},{1,1} };

/* growable int64 vector */
typedef struct { int64_t *a; int64_t n, cap; } vec;
static int vec_push(vec *v, int64_t x)
{
    if (v->n == v->cap) {
        int64_t nc = v->cap ? v->cap * 2 : 16;
        int64_t *na = (int64_t *)realloc(v->a, sizeof(int64_t) * nc);
        if (!na) return -2;
        v->a = na; v->cap = nc;
    }
    v->a[v->n++] = x;
    return 0;
}

/* Shared by v1 and block: the unrotated grid of buildGrids (cDBSCAN.py:72-90,
 * blockDBSCAN.py:69-86): cells in insertion order, CSR member lists in row order. */
typedef struct {
    celldict d;
    int64_t *pcell;    /* cell of row i */
    int64_t *cstart;   /* CSR offsets, ncell+1 */
    int64_t *cmem;     /* rows, grouped by cell, ascending row inside a cell */
    int64_t *nb;       /* ncell*8 : neighbour cell index or -1 (order NB8) */
    uint8_t *alive;    /* 0 after removeNoiseGrids deleted the cell */
} ugrid;

static void ug_free(ugrid *g)
{
    cd_free(&g->d); free(g->pcell); free(g->cstart); free(g->cmem); free(g->nb); free(g->alive);
}

static int ug_build(ugrid *g, const int64_t *X, const int64_t *Y, int64_t n, int64_t eps, int64_t minPts)
{
    memset(g, 0, sizeof(*g));
    if (cd_init(&g->d, n)) return -2;
    g->pcell = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    if (!g->pcell) return -2;
    /* minX, minY  (cDBSCAN.py:77-80) */
    int64_t minX = X[0], minY = Y[0];
    for (int64_t i = 0; i < n; i++) { if (X[i] < minX) minX = X[i]; if (Y[i] < minY) minY = Y[i]; }
    for (int64_t i = 0; i < n; i++) {
        int64_t nx = py_int_truediv(X[i] - minX, eps) + 1;   /* :84 */
        int64_t ny = py_int_truediv(Y[i] - minY, eps) + 1;   /* :85 */
        g->pcell[i] = cd_get_or_add(&g->d, nx, ny);          /* :86-87 */
    }
    int64_t C = g->d.ncell;
    g->cstart = (int64_t *)calloc((size_t)C + 2, sizeof(int64_t));
    g->cmem = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
    g->nb = (int64_t *)malloc(sizeof(int64_t) * 8 * (C + 1));
    g->alive = (uint8_t *)malloc((size_t)C + 1);
    if (!g->cstart || !g->cmem || !g->nb || !g->alive) return -2;
    for (int64_t i = 0; i < n; i++) g->cstart[g->pcell[i] + 1]++;
    for (int64_t c = 0; c < C; c++) g->cstart[c + 1] += g->cstart[c];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (C + 1));
    if (!fill) return -2;
    memcpy(fill, g->cstart, sizeof(int64_t) * C);
    for (int64_t i = 0; i < n; i++) g->cmem[fill[g->pcell[i]]++] = i;
    free(fill);
    for (int64_t c = 0; c < C; c++) {
        g->alive[c] = 1;
        for (int k = 0; k < 8; k++)
            g->nb[c * 8 + k] = cd_find(&g->d, g->d.knx[c] + NB8[k][0], g->d.kny[c] + NB8[k][1]);
    }
    /* removeNoiseGrids (cDBSCAN.py:105-126, blockDBSCAN.py:101-122):
     * tode2 = cells whose 9-cell population (len(Gs2[cell])) < minPts;
     * tode  = tode2 cells all of whose existing neighbours are in tode2 too. */
    uint8_t *tode2 = (uint8_t *)malloc((size_t)C + 1);
    if (!tode2) return -2;
    for (int64_t c = 0; c < C; c++) {
        int64_t tot = g->cstart[c + 1] - g->cstart[c];
        for (int k = 0; k < 8; k++) {
            int64_t q = g->nb[c * 8 + k];
            if (q >= 0) tot += g->cstart[q + 1] - g->cstart[q];
        }
        tode2[c] = tot < minPts;
    }
    for (int64_t c = 0; c < C; c++) {
        if (!tode2[c]) continue;
        int all = 1;
        for (int k = 0; k < 8; k++) {
            int64_t q = g->nb[c * 8 + k];
            if (q >= 0 && !tode2[q]) { all = 0; break; }
        }
        if (all) g->alive[c] = 0;
    }
    free(tode2);
    /* second buildGridNeighbors (cDBSCAN.py:37): deleted cells vanish from neighbour lists */
    for (int64_t c = 0; c < C; c++)
        for (int k = 0; k < 8; k++) {
            int64_t q = g->nb[c * 8 + k];
            if (q >= 0 && !g->alive[q]) g->nb[c * 8 + k] = -1;
        }
    return 0;
}

/* ====================================================================================
 * v1 : cLoops/cDBSCAN.py
 * ================================================================================== */

/* regionQuery (cDBSCAN.py:186-205): [p] + every q != p of Gs2[cell(p)] (own cell first,
 * then neighbour cells in NB8 order) with city-block distance <= eps (getDist :42-51). */
static int v1_region_query(const ugrid *g, const int64_t *X, const int64_t *Y, int64_t eps,
                           int64_t p, vec *out)
{
    out->n = 0;
    if (vec_push(out, p)) return -2;
    int64_t c = g->pcell[p];
    for (int k = -1; k < 8; k++) {
        int64_t q = (k < 0) ? c : g->nb[c * 8 + k];
        if (q < 0) continue;
        for (int64_t t = g->cstart[q]; t < g->cstart[q + 1]; t++) {
            int64_t r = g->cmem[t];
            if (r == p) continue;
            int64_t dx = X[p] - X[r], dy = Y[p] - Y[r];
            if (dx < 0) dx = -dx;
            if (dy < 0) dy = -dy;
            if (dx + dy <= eps) if (vec_push(out, r)) return -2;
        }
    }
    return 0;
}

int cl_oracle_v1(const int64_t *X, const int64_t *Y, int64_t n, int64_t eps, int64_t minPts,
                 int32_t *labels)
{
    if (n <= 0) return -1;                       /* mat[0] -> IndexError (cDBSCAN.py:77) */
    ugrid g;
    int rc = ug_build(&g, X, Y, n, eps, minPts);
    if (rc) { ug_free(&g); return rc; }
    /* ps[id][-1]: -1 unclassified, -2 noise, >=0 cluster; -3 here = point deleted with its cell */
    int64_t *lab = (int64_t *)malloc(sizeof(int64_t) * n);
    if (!lab) { ug_free(&g); return -2; }
    for (int64_t i = 0; i < n; i++) lab[i] = g.alive[g.pcell[i]] ? -1 : -3;
    vec seeds = {0, 0, 0}, res = {0, 0, 0};
    int64_t clusterId = 0;
    /* callClusters (cDBSCAN.py:128-137): points in dict (= row) order */
    for (int64_t p = 0; p < n && !rc; p++) {
        if (lab[p] != -1) continue;
        /* expandCluster (cDBSCAN.py:155-184) */
        if ((rc = v1_region_query(&g, X, Y, eps, p, &seeds))) break;
        if (seeds.n < minPts) { lab[p] = -2; continue; }          /* :168-170 */
        for (int64_t t = 0; t < seeds.n; t++) lab[seeds.a[t]] = clusterId;   /* :172-173 overwrite */
        for (int64_t head = 0; head < seeds.n && !rc; head++) {   /* seeds[0] ... del seeds[0] */
            int64_t cur = seeds.a[head];
            if ((rc = v1_region_query(&g, X, Y, eps, cur, &res))) break;
            if (res.n >= minPts) {
                for (int64_t t = 0; t < res.n; t++) {
                    int64_t q = res.a[t];
                    if (lab[q] == -1 || lab[q] == -2) {           /* :179-182 */
                        if (lab[q] == -1) if ((rc = vec_push(&seeds, q))) break;
                        lab[q] = clusterId;
                    }
                }
            }
        }
        clusterId++;
    }
    if (!rc) {
        /* :139-152  labels for c != -2, then clusters with < minPts members are dropped
         * (ids keep their gaps) */
        int64_t *cnt = (int64_t *)calloc((size_t)clusterId + 1, sizeof(int64_t));
        if (!cnt) rc = -2;
        else {
            for (int64_t i = 0; i < n; i++) if (lab[i] >= 0) cnt[lab[i]]++;
            for (int64_t i = 0; i < n; i++)
                labels[i] = (lab[i] >= 0 && cnt[lab[i]] >= minPts) ? (int32_t)lab[i] : -1;
            free(cnt);
        }
    }
    free(seeds.a); free(res.a); free(lab); ug_free(&g);
    return rc;
}

/* ====================================================================================
 * block : cLoops/blockDBSCAN.py
 * ================================================================================== */
typedef struct {
    const ugrid *g; const int64_t *X, *Y; int64_t eps;
    const double *cx, *cy; const int64_t *cn;
} blk;

/* getGridDist (blockDBSCAN.py:204-213): any point pair with city-block distance <= eps */
static int blk_grid_dist(const blk *b, int64_t ca, int64_t cb)
{
    const ugrid *g = b->g;
    for (int64_t s = g->cstart[ca]; s < g->cstart[ca + 1]; s++) {
        int64_t p = g->cmem[s];
        for (int64_t t = g->cstart[cb]; t < g->cstart[cb + 1]; t++) {
            int64_t q = g->cmem[t];
            int64_t dx = b->X[p] - b->X[q], dy = b->Y[p] - b->Y[q];
            if (dx < 0) dx = -dx;
            if (dy < 0) dy = -dy;
            if (dx + dy <= b->eps) return 1;
        }
    }
    return 0;
}

/* regionQuery (blockDBSCAN.py:215-239): linked neighbour cells + population sum */
static int blk_region_query(const blk *b, int64_t c, vec *out, int64_t *psum)
{
    out->n = 0;
    if (vec_push(out, c)) return -2;
    int64_t s = b->cn[c];
    for (int k = 0; k < 8; k++) {
        int64_t q = b->g->nb[c * 8 + k];
        if (q < 0) continue;
        /* getDist on the float centroids (:232), compared with eps as float64 */
        double d = fabs(b->cx[c] - b->cx[q]) + fabs(b->cy[c] - b->cy[q]);
        if (d <= (double)b->eps || blk_grid_dist(b, c, q)) {
            if (vec_push(out, q)) return -2;
            s += b->cn[q];
        }
    }
    *psum = s;
    return 0;
}

int cl_oracle_block(const int64_t *X, const int64_t *Y, int64_t n, int64_t eps, int64_t minPts,
                    int32_t *labels)
{
    if (n <= 0) return -1;                       /* blockDBSCAN.py:74 IndexError */
    ugrid g;
    int rc = ug_build(&g, X, Y, n, eps, minPts);
    if (rc) { ug_free(&g); return rc; }
    int64_t C = g.d.ncell;
    /* centerGrids (blockDBSCAN.py:124-140): [sumX/n, sumY/n, n, -1] with true division */
    double *cx = (double *)malloc(sizeof(double) * (C + 1));
    double *cy = (double *)malloc(sizeof(double) * (C + 1));
    int64_t *cn = (int64_t *)malloc(sizeof(int64_t) * (C + 1));
    int64_t *clab = (int64_t *)malloc(sizeof(int64_t) * (C + 1));
    vec seeds = {0, 0, 0}, res = {0, 0, 0};
    if (!cx || !cy || !cn || !clab) rc = -2;
    if (!rc) {
        for (int64_t c = 0; c < C; c++) {
            int64_t sx = 0, sy = 0, m = g.cstart[c + 1] - g.cstart[c];
            for (int64_t t = g.cstart[c]; t < g.cstart[c + 1]; t++) { sx += X[g.cmem[t]]; sy += Y[g.cmem[t]]; }
            cx[c] = (double)sx / (double)m;
            cy[c] = (double)sy / (double)m;
            cn[c] = m;
            clab[c] = g.alive[c] ? -1 : -3;
        }
        blk b = { &g, X, Y, eps, cx, cy, cn };
        int64_t clusterId = 0;
        /* callClusters (:142-152): surviving cells in insertion order */
        for (int64_t c0 = 0; c0 < C && !rc; c0++) {
            if (clab[c0] != -1) continue;
            int64_t psum;
            /* expandCluster (:170-202) */
            if ((rc = blk_region_query(&b, c0, &seeds, &psum))) break;
            if (psum < minPts) { clab[c0] = -2; continue; }
            for (int64_t t = 0; t < seeds.n; t++) clab[seeds.a[t]] = clusterId;     /* :185-186 */
            for (int64_t head = 0; head < seeds.n && !rc; head++) {                 /* seeds.pop(0) */
                int64_t cur = seeds.a[head];
                if ((rc = blk_region_query(&b, cur, &res, &psum))) break;
                if (psum < minPts) continue;                                        /* :191-192 */
                if (res.n >= 2) {                                                   /* :194 */
                    for (int64_t t = 0; t < res.n; t++) {
                        int64_t q = res.a[t];
                        if (clab[q] == -1) if ((rc = vec_push(&seeds, q))) break;   /* :196-197 */
                        clab[q] = clusterId;                                        /* :198 unconditional */
                    }
                }
            }
            clusterId++;
        }
        /* getLabels (:154-168): every point of a labelled cell inherits the label */
        if (!rc)
            for (int64_t i = 0; i < n; i++) {
                int64_t l = clab[g.pcell[i]];
                labels[i] = l >= 0 ? (int32_t)l : -1;
            }
    }
    free(seeds.a); free(res.a); free(cx); free(cy); free(cn); free(clab); ug_free(&g);
    return rc;
}

/* ====================================================================================
 * v2 : cLoops/cDBSCAN2.py   (production)
 * ================================================================================== */
typedef struct {
    int64_t n, eps, minPts;
    int64_t *px, *py;       /* rotated coordinates  x = X - Y, y = X + Y  (:67-68) */
    int64_t *plab;          /* p[-1]: -1 unassigned else cluster id */
    celldict d;
    int64_t *pcell;
    int64_t *cstart;        /* CSR of Grid[cell] */
    int64_t *ox;            /* members sorted by x (stable)           = Gorder['x'][cell] = Grid[cell] (:104) */
    int64_t *oy;            /* the x-sorted list stably sorted by y   = Gorder['y'][cell]               (:105) */
    int64_t *nb;            /* ncell*8, NB8 order, -1 if absent (noise cells removed, :107-109) */
    int8_t *gtype;          /* 1 crowded, 0 sparse, -1 edge, 2 core cell; -9 = deleted noise cell */
    /* border_pts: dict cell -> list of points */
    vec *bp;                /* per cell list */
    uint8_t *bp_in;         /* cell currently a key of border_pts */
    vec heap;               /* binary min-heap of active cells ordered by (nx,ny): sorted(keys)[0] (:143) */
    /* scratch */
    int64_t *stamp; int64_t stamp_gen;   /* id-membership marks (pre_ids / seedPtIds sets) */
} v2s;

static int v2_key_less(const v2s *s, int64_t a, int64_t b)
{
    if (s->d.knx[a] != s->d.knx[b]) return s->d.knx[a] < s->d.knx[b];
    return s->d.kny[a] < s->d.kny[b];
}
static int heap_push(v2s *s, int64_t c)
{
    if (vec_push(&s->heap, c)) return -2;
    int64_t i = s->heap.n - 1;
    while (i > 0) {
        int64_t p = (i - 1) / 2;
        if (!v2_key_less(s, s->heap.a[i], s->heap.a[p])) break;
        int64_t t = s->heap.a[i]; s->heap.a[i] = s->heap.a[p]; s->heap.a[p] = t; i = p;
    }
    return 0;
}
static int64_t heap_pop(v2s *s)
{
    int64_t top = s->heap.a[0];
    s->heap.a[0] = s->heap.a[--s->heap.n];
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < s->heap.n && v2_key_less(s, s->heap.a[l], s->heap.a[m])) m = l;
        if (r < s->heap.n && v2_key_less(s, s->heap.a[r], s->heap.a[m])) m = r;
        if (m == i) break;
        int64_t t = s->heap.a[i]; s->heap.a[i] = s->heap.a[m]; s->heap.a[m] = t; i = m;
    }
    return top;
}

/* sort helpers: stable insertion/merge sort of an index list by a coordinate array */
static void stable_sort_by(int64_t *idx, int64_t m, const int64_t *key, int64_t *tmp)
{
    if (m < 2) return;
    if (m <= 16) {
        for (int64_t i = 1; i < m; i++) {
            int64_t v = idx[i], j = i;
            while (j > 0 && key[idx[j - 1]] > key[v]) { idx[j] = idx[j - 1]; j--; }
            idx[j] = v;
        }
        return;
    }
    int64_t h = m / 2;
    stable_sort_by(idx, h, key, tmp);
    stable_sort_by(idx + h, m - h, key, tmp);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < m) tmp[k++] = (key[idx[j]] < key[idx[i]]) ? idx[j++] : idx[i++];
    while (i < h) tmp[k++] = idx[i++];
    while (j < m) tmp[k++] = idx[j++];
    memcpy(idx, tmp, sizeof(int64_t) * m);
}

/* bisect on an ordered member list; returns the sub-range [lo,hi) that binSearchAdjPt
 * (cDBSCAN2.py:364-378) slices out:
 *   delta=+1 : pts[0:bisect_right(pos, q+eps)]   (coords <= q + eps)
 *   delta=-1 : pts[bisect_left(pos, q-eps):]     (coords >= q - eps)            */
static void v2_bin_range(const v2s *s, int64_t cell, const int64_t *order, const int64_t *coord,
                         int64_t qpos, int delta, int64_t *lo, int64_t *hi)
{
    int64_t b = s->cstart[cell], e = s->cstart[cell + 1];
    int64_t xpos = qpos + s->eps * delta;
    if (delta == 1) {            /* bisect_right */
        int64_t l = b, r = e;
        while (l < r) { int64_t m = (l + r) / 2; if (xpos < coord[order[m]]) r = m; else l = m + 1; }
        *lo = b; *hi = l;
    } else {                     /* bisect_left */
        int64_t l = b, r = e;
        while (l < r) { int64_t m = (l + r) / 2; if (coord[order[m]] < xpos) l = m + 1; else r = m; }
        *lo = l; *hi = e;
    }
}

/* Points of neighbour cell `nc` (at offset dx,dy from the query's cell) adjacent to query
 * point q, as computed at cDBSCAN2.py:320-332 / :223-225:
 *   dy == 0 : one bisect on x;  dx == 0 : one bisect on y;
 *   else overlapPtList(bisect on x, bisect on y)  (:380-383, intersection by id).
 * The intersection is evaluated as "x-range of the x-order, filtered by the y predicate"
 * -- the same id set; its element order (Python set order in the reference) never
 * influences a label.  Appends to `out`. */
static int v2_adjacent(const v2s *s, int64_t nc, int dx, int dy, int64_t q, vec *out)
{
    int64_t lo, hi;
    if (dy == 0) {
        v2_bin_range(s, nc, s->ox, s->px, s->px[q], dx, &lo, &hi);
        for (int64_t t = lo; t < hi; t++) if (vec_push(out, s->ox[t])) return -2;
    } else if (dx == 0) {
        v2_bin_range(s, nc, s->oy, s->py, s->py[q], dy, &lo, &hi);
        for (int64_t t = lo; t < hi; t++) if (vec_push(out, s->oy[t])) return -2;
    } else {
        v2_bin_range(s, nc, s->ox, s->px, s->px[q], dx, &lo, &hi);
        int64_t ypos = s->py[q] + s->eps * dy;
        for (int64_t t = lo; t < hi; t++) {
            int64_t r = s->ox[t];
            if (dy == 1 ? (s->py[r] <= ypos) : (s->py[r] >= ypos)) if (vec_push(out, r)) return -2;
        }
    }
    return 0;
}

/* updatePtDict(border_pts, {cell: pts}) for one cell (cDBSCAN2.py:348-362): append the
 * points not yet listed for that cell; with checkPt keep only unassigned points. */
static int v2_bp_merge(v2s *s, int64_t cell, const int64_t *pts, int64_t m, int checkPt)
{
    vec *L = &s->bp[cell];
    int64_t gen = ++s->stamp_gen;
    if (s->bp_in[cell]) for (int64_t t = 0; t < L->n; t++) s->stamp[L->a[t]] = gen;
    int any = 0;
    for (int64_t t = 0; t < m; t++) {
        int64_t p = pts[t];
        if (checkPt && s->plab[p] != -1) continue;
        any = 1;
        if (s->stamp[p] == gen) continue;
        s->stamp[p] = gen;
        if (!s->bp_in[cell]) { L->n = 0; s->bp_in[cell] = 1; if (heap_push(s, cell)) return -2; }
        if (vec_push(L, p)) return -2;
    }
    (void)any;
    return 0;
}

/* a {cell: [points]} dict built by the neighbour searches, kept as 8 slots (one per
 * neighbour direction) of the current cell */
typedef struct { vec pts[8]; uint8_t present[8]; } adj8;
static void adj8_clear(adj8 *a) { for (int k = 0; k < 8; k++) { a->pts[k].n = 0; a->present[k] = 0; } }
static void adj8_free(adj8 *a) { for (int k = 0; k < 8; k++) free(a->pts[k].a); }

/* getSparseCellNeighbor (cDBSCAN2.py:304-346).  `seed`/`nseed` = seedpts.  Result: `tot`
 * (totalresult: neighbour direction -> unassigned adjacent points of the core points
 * found) and *flag.  */
static int v2_sparse_nb(v2s *s, int64_t cell, const int64_t *seed, int64_t nseed,
                        adj8 *tot, int *flag, vec *work, adj8 *padj)
{
    int64_t cell_pt_num = s->cstart[cell + 1] - s->cstart[cell];
    adj8_clear(tot);
    *flag = 0;
    work->n = 0;
    for (int64_t t = 0; t < nseed; t++) if (vec_push(work, seed[t])) return -2;   /* pts = seedpts[:] */
    while (work->n > 0) {
        int64_t p = work->a[--work->n];                                           /* pts.pop() */
        adj8_clear(padj);
        int64_t ncount = 0;
        for (int k = 0; k < 8; k++) {
            int64_t nc = s->nb[cell * 8 + k];
            if (nc < 0) continue;
            if (v2_adjacent(s, nc, NB8[k][0], NB8[k][1], p, &padj->pts[k])) return -2;
            padj->present[k] = 1;
            ncount += padj->pts[k].n;
        }
        if (ncount + cell_pt_num >= s->minPts) {                                  /* :334 */
            /* updatePtDict(totalresult, p_adjacent, checkPt=True) */
            for (int k = 0; k < 8; k++) {
                if (!padj->present[k]) continue;
                int64_t gen = ++s->stamp_gen;
                for (int64_t t = 0; t < tot->pts[k].n; t++) s->stamp[tot->pts[k].a[t]] = gen;
                for (int64_t t = 0; t < padj->pts[k].n; t++) {
                    int64_t q = padj->pts[k].a[t];
                    if (s->plab[q] != -1 || s->stamp[q] == gen) continue;
                    s->stamp[q] = gen;
                    if (vec_push(&tot->pts[k], q)) return -2;
                    tot->present[k] = 1;
                }
            }
            if (!*flag) {                                                         /* :337-345 */
                int64_t gen = ++s->stamp_gen;
                for (int64_t t = 0; t < nseed; t++) s->stamp[seed[t]] = gen;
                for (int64_t t = s->cstart[cell]; t < s->cstart[cell + 1]; t++) {
                    int64_t q = s->ox[t];
                    if (s->plab[q] == -1 && s->stamp[q] != gen) if (vec_push(work, q)) return -2;
                }
                *flag = 1;
            }
        }
    }
    return 0;
}

/* findEdgePts (cDBSCAN2.py:244-302): the four Pareto staircases of a crowded cell, walked
 * over the x-order with the reference's early-exit flags.  stair[0..3] = (-1,-1) downleft,
 * (-1,1) upleft, (1,-1) downright, (1,1) upright. */
static int v2_find_edge_pts(const v2s *s, int64_t cell, vec stair[4])
{
    int64_t b = s->cstart[cell], e = s->cstart[cell + 1];
    int64_t ymax = s->py[s->oy[e - 1]], ymin = s->py[s->oy[b]];
    for (int k = 0; k < 4; k++) stair[k].n = 0;
    vec *downleft = &stair[0], *upleft = &stair[1], *downright = &stair[2], *upright = &stair[3];
    if (vec_push(upleft, s->ox[b]) || vec_push(downleft, s->ox[b])) return -2;
    int up = 1, down = 1;
    for (int64_t t = b + 1; t < e; t++) {                       /* order['x'][1:] */
        int64_t i = s->ox[t];
        if (up) {
            int64_t j = upleft->a[upleft->n - 1];
            if (s->py[i] > s->py[j]) {
                if (s->px[i] == s->px[j]) upleft->a[upleft->n - 1] = i;
                else if (vec_push(upleft, i)) return -2;
            }
            if (s->py[i] == ymax) up = 0;
        }
        if (down) {
            int64_t j = downleft->a[downleft->n - 1];
            if (s->py[i] < s->py[j]) {
                if (s->px[i] == s->px[j]) downleft->a[downleft->n - 1] = i;
                else if (vec_push(downleft, i)) return -2;
            }
            if (s->py[i] == ymin) down = 0;
        }
        if (!(up || down)) break;
    }
    if (vec_push(upright, s->ox[e - 1]) || vec_push(downright, s->ox[e - 1])) return -2;
    up = 1; down = 1;
    for (int64_t t = e - 1; t >= b; t--) {                      /* order['x'][-1::-1] (starts at the last itself) */
        int64_t i = s->ox[t];
        if (up) {
            int64_t j = upright->a[upright->n - 1];
            if (s->py[i] > s->py[j]) {
                if (s->px[i] == s->px[j]) upright->a[upright->n - 1] = i;
                else if (vec_push(upright, i)) return -2;
            }
            if (s->py[i] == ymax) up = 0;
        }
        if (down) {
            int64_t j = downright->a[downright->n - 1];
            if (s->py[i] < s->py[j]) {
                if (s->px[i] == s->px[j]) downright->a[downright->n - 1] = i;
                else if (vec_push(downright, i)) return -2;
            }
            if (s->py[i] == ymin) down = 0;
        }
        if (!(up || down)) break;
    }
    return 0;
}

static int nb_slot(int dx, int dy)
{
    for (int k = 0; k < 8; k++) if (NB8[k][0] == dx && NB8[k][1] == dy) return k;
    return -1;
}

/* getCrowdedCellNeighbor (cDBSCAN2.py:194-242) -> adj (direction -> points).
 * adj->present[k] == 2 marks "the whole neighbour cell" (:226-229). */
static int v2_crowded_nb(v2s *s, int64_t cell, adj8 *adj, vec stair[4], vec *tmp)
{
    adj8_clear(adj);
    int64_t b = s->cstart[cell], e = s->cstart[cell + 1];
    /* axis neighbours (:196-214): extreme point along the axis + one bisect */
    for (int axis = 0; axis < 2; axis++)
        for (int delta = -1; delta <= 1; delta += 2) {
            int dx = axis == 0 ? delta : 0, dy = axis == 0 ? 0 : delta;
            int k = nb_slot(dx, dy);
            int64_t nc = s->nb[cell * 8 + k];
            if (nc < 0 || s->gtype[nc] == 2) continue;
            const int64_t *ord = axis == 0 ? s->ox : s->oy;
            int64_t edgept = delta == -1 ? ord[b] : ord[e - 1];
            tmp->n = 0;
            if (v2_adjacent(s, nc, dx, dy, edgept, tmp)) return -2;
            for (int64_t t = 0; t < tmp->n; t++)
                if (s->plab[tmp->a[t]] == -1) { if (vec_push(&adj->pts[k], tmp->a[t])) return -2; adj->present[k] = 1; }
        }
    /* diagonal neighbours (:216-241): staircase points, intersection of two bisects */
    if (v2_find_edge_pts(s, cell, stair)) return -2;
    static const int DD[4][2] = { {-1,-1}, {-1,1}, {1,-1}, {1,1} };
    for (int dd = 0; dd < 4; dd++) {
        int dx = DD[dd][0], dy = DD[dd][1];
        int k = nb_slot(dx, dy);
        int64_t nc = s->nb[cell * 8 + k];
        if (nc < 0 || s->gtype[nc] == 2) continue;
        for (int64_t u = 0; u < stair[dd].n; u++) {
            int64_t p = stair[dd].a[u];
            tmp->n = 0;
            if (v2_adjacent(s, nc, dx, dy, p, tmp)) return -2;
            if (s->gtype[nc] == 1 && tmp->n > 0) {             /* one hit equals all hit */
                adj->pts[k].n = 0;
                for (int64_t t = s->cstart[nc]; t < s->cstart[nc + 1]; t++)
                    if (vec_push(&adj->pts[k], s->ox[t])) return -2;
                adj->present[k] = 2;
                break;
            }
            /* merge the unassigned new hits, no duplicates (:230-241) */
            int64_t gen = ++s->stamp_gen;
            for (int64_t t = 0; t < adj->pts[k].n; t++) s->stamp[adj->pts[k].a[t]] = gen;
            for (int64_t t = 0; t < tmp->n; t++) {
                int64_t q = tmp->a[t];
                if (s->plab[q] != -1 || s->stamp[q] == gen) continue;
                s->stamp[q] = gen;
                if (vec_push(&adj->pts[k], q)) return -2;
                adj->present[k] = 1;
            }
        }
    }
    return 0;
}

int cl_oracle_v2(const int64_t *X, const int64_t *Y, int64_t n, int64_t eps, int64_t minPts,
                 int32_t *labels)
{
    if (n < 0) return -1;
    for (int64_t i = 0; i < n; i++) labels[i] = -1;
    if (n == 0) return 0;                        /* v2 on an empty mat yields {} */
    v2s S; memset(&S, 0, sizeof(S));
    v2s *s = &S;
    int rc = 0;
    s->n = n; s->eps = eps; s->minPts = minPts;
    s->px = (int64_t *)malloc(sizeof(int64_t) * n);
    s->py = (int64_t *)malloc(sizeof(int64_t) * n);
    s->plab = (int64_t *)malloc(sizeof(int64_t) * n);
    s->pcell = (int64_t *)malloc(sizeof(int64_t) * n);
    s->ox = (int64_t *)malloc(sizeof(int64_t) * n);
    s->oy = (int64_t *)malloc(sizeof(int64_t) * n);
    s->stamp = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * n);
    vec work = {0,0,0}, tv = {0,0,0}, cl = {0,0,0}, seedv = {0,0,0};
    vec stair[4]; memset(stair, 0, sizeof(stair));
    adj8 tot, padj, cadj; memset(&tot, 0, sizeof(tot)); memset(&padj, 0, sizeof(padj)); memset(&cadj, 0, sizeof(cadj));
    if (!s->px || !s->py || !s->plab || !s->pcell || !s->ox || !s->oy || !s->stamp || !tmp) { rc = -2; goto done; }
    if (cd_init(&s->d, n)) { rc = -2; goto done; }
    /* buildGrid (cDBSCAN2.py:55-112) */
    for (int64_t i = 0; i < n; i++) {
        s->px[i] = X[i] - Y[i];                                  /* :67 */
        s->py[i] = X[i] + Y[i];                                  /* :68 */
        int64_t nx = py_int_truediv(s->px[i], eps) + 1;          /* :69 */
        int64_t ny = py_int_truediv(s->py[i], eps) + 1;          /* :70 */
        s->pcell[i] = cd_get_or_add(&s->d, nx, ny);
        s->plab[i] = -1;
    }
    int64_t C = s->d.ncell;
    s->cstart = (int64_t *)calloc((size_t)C + 2, sizeof(int64_t));
    s->nb = (int64_t *)malloc(sizeof(int64_t) * 8 * (C + 1));
    s->gtype = (int8_t *)malloc((size_t)C + 1);
    s->bp = (vec *)calloc((size_t)C + 1, sizeof(vec));
    s->bp_in = (uint8_t *)calloc((size_t)C + 1, 1);
    if (!s->cstart || !s->nb || !s->gtype || !s->bp || !s->bp_in) { rc = -2; goto done; }
    for (int64_t i = 0; i < n; i++) s->cstart[s->pcell[i] + 1]++;
    for (int64_t c = 0; c < C; c++) s->cstart[c + 1] += s->cstart[c];
    {
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (C + 1));
        if (!fill) { rc = -2; goto done; }
        memcpy(fill, s->cstart, sizeof(int64_t) * C);
        for (int64_t i = 0; i < n; i++) s->ox[fill[s->pcell[i]]++] = i;     /* insertion (row) order */
        free(fill);
    }
    for (int64_t c = 0; c < C; c++)
        for (int k = 0; k < 8; k++)
            s->nb[c * 8 + k] = cd_find(&s->d, s->d.knx[c] + NB8[k][0], s->d.kny[c] + NB8[k][1]);
    /* cell types (:77-91) */
    for (int64_t c = 0; c < C; c++) {
        int64_t m = s->cstart[c + 1] - s->cstart[c];
        if (m >= minPts) { s->gtype[c] = 1; continue; }
        for (int k = 0; k < 8; k++) {
            int64_t q = s->nb[c * 8 + k];
            if (q >= 0) m += s->cstart[q + 1] - s->cstart[q];
        }
        s->gtype[c] = m < minPts ? -1 : 0;
    }
    /* noise cells (:93-109) and the two per-cell orders (:104-105) */
    {
        uint8_t *noise = (uint8_t *)calloc((size_t)C + 1, 1);
        if (!noise) { rc = -2; goto done; }
        for (int64_t c = 0; c < C; c++) {
            int allneg = 1;
            for (int k = 0; k < 8; k++) {
                int64_t q = s->nb[c * 8 + k];
                if (q >= 0 && s->gtype[q] != -1) { allneg = 0; break; }
            }
            if (s->gtype[c] == -1 && allneg) { noise[c] = 1; continue; }
            int64_t b = s->cstart[c], m = s->cstart[c + 1] - b;
            stable_sort_by(s->ox + b, m, s->px, tmp);
            memcpy(s->oy + b, s->ox + b, sizeof(int64_t) * m);
            stable_sort_by(s->oy + b, m, s->py, tmp);
        }
        for (int64_t c = 0; c < C; c++) if (noise[c]) s->gtype[c] = -9;
        for (int64_t c = 0; c < C; c++)
            for (int k = 0; k < 8; k++) {
                int64_t q = s->nb[c * 8 + k];
                if (q >= 0 && s->gtype[q] == -9) s->nb[c * 8 + k] = -1;
            }
        free(noise);
    }
    /* queryGrid (:114-192) */
    {
        int64_t clusterId = 0;
        for (int64_t c0 = 0; c0 < C; c0++) {
            int8_t t0 = s->gtype[c0];
            if (t0 == -9 || t0 == -1 || t0 == 2) continue;                  /* :119 */
            cl.n = 0;                                                       /* clusters[clusterId] = [] */
            if (t0 == 1) {
                /* border_pts[index] = Grid[index] (:126) */
                if (v2_bp_merge(s, c0, s->ox + s->cstart[c0], s->cstart[c0 + 1] - s->cstart[c0], 0)) { rc = -2; goto done; }
            } else {
                seedv.n = 0;                                                /* :130 unassigned points of the cell */
                for (int64_t t = s->cstart[c0]; t < s->cstart[c0 + 1]; t++)
                    if (s->plab[s->ox[t]] == -1) if (vec_push(&seedv, s->ox[t])) { rc = -2; goto done; }
                int flag;
                if (v2_sparse_nb(s, c0, seedv.a, seedv.n, &tot, &flag, &work, &padj)) { rc = -2; goto done; }
                if (!flag) continue;                                        /* :138-140 */
                for (int64_t t = 0; t < seedv.n; t++) {                     /* :134-136 */
                    s->plab[seedv.a[t]] = clusterId;
                    if (vec_push(&cl, seedv.a[t])) { rc = -2; goto done; }
                }
                for (int k = 0; k < 8; k++)                                 /* border_pts = adjacent_pts */
                    if (tot.present[k])
                        if (v2_bp_merge(s, s->nb[c0 * 8 + k], tot.pts[k].a, tot.pts[k].n, 0)) { rc = -2; goto done; }
            }
            while (s->heap.n > 0) {                                         /* :142 */
                int64_t nc = heap_pop(s);                                   /* sorted(border_pts.keys())[0] */
                int8_t ty = s->gtype[nc];
                vec *L = &s->bp[nc];
                if (ty == 1) {                                              /* :147-154 crowded cell */
                    s->gtype[nc] = 2;
                    for (int64_t t = s->cstart[nc]; t < s->cstart[nc + 1]; t++) {
                        s->plab[s->ox[t]] = clusterId;
                        if (vec_push(&cl, s->ox[t])) { rc = -2; goto done; }
                    }
                    if (v2_crowded_nb(s, nc, &cadj, stair, &tv)) { rc = -2; goto done; }
                    for (int k = 0; k < 8; k++)
                        if (cadj.present[k])
                            if (v2_bp_merge(s, s->nb[nc * 8 + k], cadj.pts[k].a, cadj.pts[k].n, 0)) { rc = -2; goto done; }
                } else if (ty == 0) {                                       /* :155-170 sparse cell */
                    int flag;
                    seedv.n = 0;
                    for (int64_t t = 0; t < L->n; t++) if (vec_push(&seedv, L->a[t])) { rc = -2; goto done; }
                    if (v2_sparse_nb(s, nc, seedv.a, seedv.n, &tot, &flag, &work, &padj)) { rc = -2; goto done; }
                    if (flag) {
                        for (int64_t t = s->cstart[nc]; t < s->cstart[nc + 1]; t++) {
                            int64_t p = s->ox[t];
                            if (s->plab[p] == -1) { s->plab[p] = clusterId; if (vec_push(&cl, p)) { rc = -2; goto done; } }
                        }
                        for (int k = 0; k < 8; k++)
                            if (tot.present[k])
                                if (v2_bp_merge(s, s->nb[nc * 8 + k], tot.pts[k].a, tot.pts[k].n, 0)) { rc = -2; goto done; }
                    } else {
                        for (int64_t t = 0; t < seedv.n; t++) {             /* :168-170 border points */
                            s->plab[seedv.a[t]] = clusterId;
                            if (vec_push(&cl, seedv.a[t])) { rc = -2; goto done; }
                        }
                    }
                } else {                                                    /* :171-176 edge cell */
                    for (int64_t t = 0; t < L->n; t++) {
                        s->plab[L->a[t]] = clusterId;
                        if (vec_push(&cl, L->a[t])) { rc = -2; goto done; }
                    }
                }
                s->bp_in[nc] = 0;                                           /* del border_pts[nindex] (:177) */
                L->n = 0;
            }
            if (cl.n < minPts) {                                            /* :180-183 release */
                for (int64_t t = 0; t < cl.n; t++) s->plab[cl.a[t]] = -1;
            } else {
                for (int64_t t = 0; t < cl.n; t++) labels[cl.a[t]] = (int32_t)clusterId;   /* :186-191 */
                clusterId++;
            }
        }
    }
done:
    free(s->px); free(s->py); free(s->plab); free(s->pcell); free(s->ox); free(s->oy); free(s->stamp);
    free(tmp); free(s->cstart); free(s->nb); free(s->gtype); free(s->bp_in);
    if (s->bp) { for (int64_t c = 0; c <= s->d.ncell; c++) free(s->bp[c].a); free(s->bp); }
    free(s->heap.a);
    if (s->d.slot) cd_free(&s->d);
    free(work.a); free(tv.a); free(cl.a); free(seedv.a);
    for (int k = 0; k < 4; k++) free(stair[k].a);
    adj8_free(&tot); adj8_free(&padj); adj8_free(&cadj);
    return rc;
}

/* ------------------------------------------------------------------------------------
 * Brute-force neighbour counts  |{q : |Xp-Xq| + |Yp-Yq| <= eps}|  (self included), the
 * quantity the reference tests against minPts (cDBSCAN.py:168,177; cDBSCAN2.py:333-334).
 * O(n^2): for small parity cases of the region-query kernel only.
 * ---------------------------------------------------------------------------------- */
int cl_oracle_neighbor_counts(const int64_t *X, const int64_t *Y, int64_t n, int64_t eps, int32_t *cnt)
{
    for (int64_t i = 0; i < n; i++) {
        int32_t c = 0;
        for (int64_t j = 0; j < n; j++) {
            int64_t dx = X[i] - X[j], dy = Y[i] - Y[j];
            if (dx < 0) dx = -dx;
            if (dy < 0) dy = -dy;
            if (dx + dy <= eps) c++;
        }
        cnt[i] = c;
    }
    return 0;
}

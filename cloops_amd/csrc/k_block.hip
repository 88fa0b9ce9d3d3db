// k_block.hip -- K6 of libcloops_hip.so: blockDBSCAN (cLoops/blockDBSCAN.py:6-239), kernels and driver.
#include "cl_chrom.h"

// ==========================================================================================
// K6: blockDBSCAN (cLoops/blockDBSCAN.py) -- DBSCAN over grid CELLS
// ==========================================================================================
// Closed form: unrotated eps-grid anchored at the filtered set's (minX,minY)
// (:74-82); cells whose 9-cell population is < minPts and whose existing neighbours are all
// like that are deleted (:101-122); two surviving 8-adjacent cells are LINKED iff their
// float64 centroids are within eps (city block) or some point pair is (:204-239); a cell is
// CORE iff own + linked population >= minPts (:181,191); components of core cells are ranked
// by their first cell in insertion order (= smallest input row of the cell's first point,
// :148-152); a non-core cell linked to core cells takes the LARGEST adjacent rank
// (unconditional overwrite, :195-198); points inherit their cell's label (:154-168).
struct BlkParams {
    int eps, minPts, cut;
    int R;            // rows of the cell-row table; key row R marks filtered PETs
    int n;
    int nyb, rb;      // sort key = ((nx << nyb | ny) << 2*rb) | rx << rb | ry ; only the cell bits are sorted
    u32 magic; int sh1, sh2;      // v / eps by multiply-shift (same constants as GridParams)
};
__device__ __forceinline__ u32 blk_div(const BlkParams& p, u32 n)
{
    const u32 t1 = __umulhi(p.magic, n);
    return (t1 + ((n - t1) >> p.sh1)) >> p.sh2;
}
struct BlkScalars { int minx, miny, M, C; };

__global__ void k_blk_init_scalars(BlkScalars* sc, int minx, int miny)
{
    sc->minx = minx; sc->miny = miny; sc->M = 0; sc->C = 0;
}

__global__ void k_blk_minmax(const int* __restrict__ X, const int* __restrict__ Y, int n, int cut, BlkScalars* sc)
{
    __shared__ int red[2][TPB / 64];
    int mx = INT_MAX, my = INT_MAX;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int x = X[i], y = Y[i];
        if (y - x < cut) continue;
        mx = min(mx, x); my = min(my, y);
    }
    for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_down(mx, o)); my = min(my, __shfl_down(my, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx; red[1][threadIdx.x >> 6] = my; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) { mx = min(mx, red[0][w]); my = min(my, red[1][w]); }
        if (mx != INT_MAX) { atomicMin(&sc->minx, mx); atomicMin(&sc->miny, my); }
    }
}

__global__ void k_blk_keys(const int* __restrict__ X, const int* __restrict__ Y, BlkParams p, const BlkScalars* __restrict__ sc,
                           u64* __restrict__ keys, u32* __restrict__ vals)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n) return;
    int x = X[r], y = Y[r];
    bool valid = (p.cut <= 0) || (y - x >= p.cut);
    // nx = int((X - minX) / cw) + 1 (:81-82); the +1 is dropped (cells are only compared)
    u64 key = (u64)(u32)p.R << (p.nyb + 2 * p.rb);
    if (valid) {
        const u32 ux = (u32)(x - sc->minx), uy = (u32)(y - sc->miny);
        const u32 nx = blk_div(p, ux), ny = blk_div(p, uy);
        const u32 rx = ux - nx * (u32)p.eps, ry = uy - ny * (u32)p.eps;
        key = ((((u64)nx << p.nyb) | ny) << (2 * p.rb)) | ((u64)rx << p.rb) | ry;
    }
    keys[r] = key;
    vals[r] = (u32)r;
}

// decode the sorted keys back into coordinates (no gather), mark cell heads
__global__ void k_blk_gather(BlkParams p, const u64* __restrict__ skeys,
                             int* __restrict__ sx, int* __restrict__ sy, int* __restrict__ headflag, BlkScalars* sc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const int cb = 2 * p.rb;
    const u64 k = skeys[i];
    const u64 cell = k >> cb;
    const u32 nx = (u32)(cell >> p.nyb), ny = (u32)(cell & ((1ull << p.nyb) - 1));
    const bool valid = nx < (u32)p.R;
    const u32 rmask = (1u << p.rb) - 1;
    sx[i] = sc->minx + (int)(nx * (u32)p.eps + ((u32)(k >> p.rb) & rmask));
    sy[i] = sc->miny + (int)(ny * (u32)p.eps + ((u32)k & rmask));
    const u64 prev = i ? (skeys[i - 1] >> cb) : ~0ull;
    headflag[i] = (valid && prev != cell) ? 1 : 0;
    if (!valid && (i == 0 || (u32)(prev >> p.nyb) < (u32)p.R)) sc->M = i;   // first filtered row
    if (valid && i == p.n - 1) sc->M = p.n;
}

// cid[i] = (inclusive prefix sum of headflag)[i] - 1 ; per-cell arrays
__global__ void k_blk_cells(BlkParams p, const u64* __restrict__ skeys, const int* __restrict__ headflag,
                            const int* __restrict__ cidp1, const u32* __restrict__ srow, BlkScalars* sc,
                            int* __restrict__ cstart, u64* __restrict__ ckey, int* __restrict__ cfirst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = sc->M;
    if (i >= M) return;
    if (headflag[i]) {
        int c = cidp1[i] - 1;
        cstart[c] = i;
        ckey[c] = skeys[i] >> (2 * p.rb);
        cfirst[c] = (int)srow[i];       // stable sort: first of the run = smallest input row
    }
    if (i == M - 1) { int C = cidp1[i]; sc->C = C; cstart[C] = M; }
}

// rowcell[r] = first cell index whose row >= r, r = 0..R
__global__ void k_blk_rowtable(BlkParams p, const BlkScalars* __restrict__ sc, const u64* __restrict__ ckey,
                               int* __restrict__ rowcell)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > p.R) return;
    const int C = sc->C;
    u64 target = (u64)(u32)r << p.nyb;
    int lo = 0, hi = C;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (ckey[mid] < target) lo = mid + 1; else hi = mid; }
    rowcell[r] = lo;
}

// neighbour indices (8 per cell, order (dx,dy) = (-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1); the
// reverse of direction q is 7-q), 9-cell population test, centroids.  Cells of one row are consecutive in
// the cell table, so each neighbouring row costs ONE binary search (for ny-1) plus a walk over <= 3 cells.
__global__ void k_blk_neighbors(BlkParams p, const BlkScalars* __restrict__ sc, const u64* __restrict__ ckey,
                                const int* __restrict__ rowcell, const int* __restrict__ cstart,
                                const int* __restrict__ sx, const int* __restrict__ sy,
                                int* __restrict__ nb, int* __restrict__ low, double* __restrict__ cx, double* __restrict__ cy)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = sc->C;
    if (c >= C) return;
    const u64 k = ckey[c];
    const long long nx = (long long)(k >> p.nyb), ny = (long long)(k & ((1ull << p.nyb) - 1));
    const int cb = cstart[c], ce = cstart[c + 1];
    int tot = ce - cb;
    int res[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) res[q] = -1;
    // same row: the neighbours are c-1 / c+1 when their keys are k-1 / k+1 (ny-1 >= 0 checked: k-1 would borrow)
    if (ny > 0 && c > 0 && ckey[c - 1] == k - 1) res[3] = c - 1;
    if (c + 1 < C && ckey[c + 1] == k + 1 && ny + 1 < (1ll << p.nyb)) res[4] = c + 1;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const long long rx = nx + (side ? 1 : -1);
        if (rx < 0 || rx >= p.R) continue;
        const int rlo = rowcell[rx], rhi = rowcell[rx + 1];
        const long long y0 = ny > 0 ? ny - 1 : 0;
        const u64 target = ((u64)rx << p.nyb) | (u64)y0;
        int lo = rlo, hi = rhi;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ckey[mid] < target) lo = mid + 1; else hi = mid; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (lo >= rhi) break;
            const long long yy = (long long)(ckey[lo] & ((1ull << p.nyb) - 1));
            const long long d = yy - ny;
            if (d > 1) break;
            if (d == -1) res[side * 5 + 0] = lo;       // side 0 -> q 0..2, side 1 -> q 5..7
            if (d == 0) res[side * 5 + 1] = lo;
            if (d == 1) res[side * 5 + 2] = lo;
            ++lo;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) if (res[q] >= 0) tot += cstart[res[q] + 1] - cstart[res[q]];
    int4* o = reinterpret_cast<int4*>(nb + (size_t)c * 8);
    o[0] = make_int4(res[0], res[1], res[2], res[3]);
    o[1] = make_int4(res[4], res[5], res[6], res[7]);
    low[c] = tot < p.minPts ? 1 : 0;
    long long sumx = 0, sumy = 0;
    for (int t = cb; t < ce; ++t) { sumx += sx[t]; sumy += sy[t]; }
    double m = (double)(ce - cb);
    cx[c] = (double)sumx / m;          // true division of Python ints (:136-137)
    cy[c] = (double)sumy / m;
}

// alive[c] = population of the cell if it survives the 9-cell test (:215-228), else 0
__global__ void k_blk_alive(const BlkScalars* __restrict__ sc, const int* __restrict__ nb, const int* __restrict__ low,
                            const int* __restrict__ cstart, int* __restrict__ alive, int* __restrict__ linkbits)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int a = 1;
    if (low[c]) {
        a = 0;
        for (int q = 0; q < 8; ++q) { int j = nb[(size_t)c * 8 + q]; if (j >= 0 && !low[j]) { a = 1; break; } }
    }
    alive[c] = a ? cstart[c + 1] - cstart[c] : 0;
    linkbits[c] = 0;
}

// link bits: one thread per (cell, forward direction q = 4..7); the link test is symmetric (same centroid
// distance, same point pairs), so the thread sets bit q of its cell and bit 7-q of the neighbour.
// (A work list + 16 lanes per undecided cell pair was measured slower: the list append costs more
// than the pair loops it spreads.)
__global__ void k_blk_links(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                            const int* __restrict__ alive, const int* __restrict__ cstart,
                            const int* __restrict__ sx, const int* __restrict__ sy,
                            const double* __restrict__ cx, const double* __restrict__ cy,
                            int* __restrict__ linkbits)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = t >> 2, q = 4 + (t & 3);
    if (c >= sc->C) return;
    const int j = nb[(size_t)c * 8 + q];
    if (j < 0) return;
    const int na = alive[c], nbj = alive[j];
    if (!na || !nbj) return;
    bool linked = (fabs(cx[c] - cx[j]) + fabs(cy[c] - cy[j])) <= (double)p.eps;        // :232
    // two single-PET cells: the centroids ARE the PETs, the pair test below cannot differ
    if (!linked && (na > 1 || nbj > 1)) {                                              // getGridDist :204-213
        int ab = cstart[c], ae = ab + na, bb = cstart[j], be = bb + nbj;
        if (na > nbj) { int x0 = ab, x1 = ae; ab = bb; ae = be; bb = x0; be = x1; }    // walk the larger cell inside
        for (int s = ab; s < ae && !linked; ++s) {
            const int x = sx[s], y = sy[s];
            int u = bb;
            for (; u + 4 <= be && !linked; u += 4) {
                int d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int dx = x - sx[u + k], dy = y - sy[u + k];
                    d[k] = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
                }
                linked = min(min(d[0], d[1]), min(d[2], d[3])) <= p.eps;
            }
            for (; u < be && !linked; ++u) {
                int dx = x - sx[u], dy = y - sy[u];
                linked = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy) <= p.eps;
            }
        }
    }
    if (linked) { atomicOr(&linkbits[c], 1 << q); atomicOr(&linkbits[j], 1 << (7 - q)); }
}

// population over the linked cells + core flag (:236-240)
__global__ void k_blk_core(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                           const int* __restrict__ alive, const int* __restrict__ cstart,
                           const int* __restrict__ linkbits, int* __restrict__ corec)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int core = 0;
    if (alive[c]) {
        const int bits = linkbits[c];
        int psum = cstart[c + 1] - cstart[c];
        for (int q = 0; q < 8; ++q)
            if (bits & (1 << q)) psum += je_minus(cstart, nb[(size_t)c * 8 + q]);
        core = psum >= p.minPts ? 1 : 0;
    }
    corec[c] = core;
}

__global__ void k_blk_union(const BlkScalars* __restrict__ sc, const int* __restrict__ nb, const int* __restrict__ linkbits,
                            const int* __restrict__ corec, int* parent)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    if (!corec[c]) return;
    int bits = linkbits[c];
    for (int q = 0; q < 8; ++q) {
        if (!(bits & (1 << q))) continue;
        int j = nb[(size_t)c * 8 + q];
        if (j < c && corec[j]) uf_unite(parent, c, j);
    }
}

// root per core cell + component key = smallest cfirst (two-level reduce-by-key like k_flatten: a giant
// component would otherwise serialise millions of atomicMin on one address)
__global__ void __launch_bounds__(BIGTPB)
k_blk_flatten(const BlkScalars* __restrict__ sc, const int* __restrict__ corec, int* parent,
              const int* __restrict__ cfirst, int* __restrict__ root, int* __restrict__ compkey)
{
    __shared__ int hkey[AGG_H], hmin[AGG_H];
    if (threadIdx.x < AGG_H) { hkey[threadIdx.x] = -1; hmin[threadIdx.x] = INT_MAX; }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = -1, key = INT_MAX;
    if (c < sc->C) {
        if (corec[c]) { r = uf_find(parent, c); key = cfirst[c]; }
        root[c] = r;
    }
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(r >= 0);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int R = __builtin_amdgcn_readlane(r, leader);
        const unsigned long long m = __ballot(r == R);
        const bool mine = r == R;
        if (__popcll(m) >= 4) {
            int mk = mine ? key : INT_MAX;
            mk = dpp_reduce_wave(mk, OpMin());
            if (lane == leader) {
                const int sl = agg_slot(hkey, R);
                if (sl >= 0) atomicMin(&hmin[sl], mk); else atomicMin(&compkey[R], mk);
            }
        } else if (mine) {
            atomicMin(&compkey[R], key);
        }
        pending &= ~m;
    }
    __syncthreads();
    if (threadIdx.x < AGG_H && hkey[threadIdx.x] >= 0) atomicMin(&compkey[hkey[threadIdx.x]], hmin[threadIdx.x]);
}

__global__ void k_blk_rank_flags(const BlkScalars* __restrict__ sc, const int* __restrict__ root,
                                 const int* __restrict__ compkey, int* __restrict__ flag)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    if (root[c] == c) flag[compkey[c]] = 1;
}

__global__ void k_blk_cell_labels(const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                                  const int* __restrict__ linkbits, const int* __restrict__ alive,
                                  const int* __restrict__ root, const int* __restrict__ compkey,
                                  const int* __restrict__ rankscan, int* __restrict__ clab)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int lab = -1;
    if (root[c] >= 0) lab = rankscan[compkey[root[c]]];
    else if (alive[c]) {
        int bits = linkbits[c];
        for (int q = 0; q < 8; ++q) {
            if (!(bits & (1 << q))) continue;
            int j = nb[(size_t)c * 8 + q];
            if (root[j] >= 0) lab = max(lab, rankscan[compkey[root[j]]]);     // :195-198 last writer = highest rank
        }
    }
    clab[c] = lab;
}

__global__ void __launch_bounds__(BIGTPB)
k_blk_point_labels(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ cidp1,
                   const int* __restrict__ clab, const u32* __restrict__ srow, const int* __restrict__ sx,
                   const int* __restrict__ sy, int* __restrict__ labels, Table t)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int M = sc->M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lab = -1, x = 0, y = 0;
    if (i < M) {
        lab = clab[cidp1[i] - 1];
        labels[srow[i]] = lab;
        x = sx[i]; y = sy[i];
    }
    table_accumulate(t, h, lab, x, y);
    table_flush(t, h);
}


// ---- variant 3 host driver -----------------------------------------------------------------
int run_block(cl_chrom* c, int eps, int minPts, int cut, int32_t* labels_out)
{
    int rc;
    const int n = (int)c->n;
    long long R = ((long long)c->st.xmax - c->st.xmin) / eps + 1;
    if (R > (1LL << 28)) return fail(CL_ERR_GRID, "eps too small for the coordinate extent (cell-row table > 2^28 rows)");
    if ((rc = ensure_workspace(c, (int)R))) return rc;
    if ((rc = ensure_events(c))) return rc;
#define ENSB(buf, bytes) if ((rc = c->buf.ensure(bytes))) return rc
    ENSB(b_cstart, ((size_t)n + 1) * 4); ENSB(b_ckey, (size_t)n * 8); ENSB(b_nb, (size_t)n * 32);
    ENSB(b_cx, (size_t)n * 8); ENSB(b_cy, (size_t)n * 8);
#undef ENSB
    BlkParams p; p.eps = eps; p.minPts = minPts; p.cut = cut; p.R = (int)R; p.n = n;
    p.nyb = std::max(1, bits_for((unsigned)(((long long)c->st.ymax - c->st.ymin) / eps)));
    p.rb = bits_for((unsigned)(eps - 1));
    {
        const unsigned d = (unsigned)eps;
        int l = 0; while ((1ull << l) < d) ++l;
        p.magic = (u32)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
        p.sh1 = l < 1 ? l : 1; p.sh2 = l > 1 ? l - 1 : 0;
    }
    int* counters = c->counters.as<int>();
    BlkScalars* sc = (BlkScalars*)(counters + 32);
    LAUNCH(k_init_arrays, n + 1, n, c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
           c->usize.as<int>(), c->cellfirst.as<int>(), c->flag.as<int>(), c->state.as<int>(), counters);
    HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    // minX / minY of the (filtered) mat (blockDBSCAN.py:74-80): known from the upload statistics when
    // nothing is filtered, one reduction pass otherwise
    hipLaunchKernelGGL(k_blk_init_scalars, dim3(1), dim3(1), 0, c->stream, sc, cut > 0 ? INT_MAX : c->st.xmin,
                       cut > 0 ? INT_MAX : c->st.ymin);
    ev_record(c, 0);
    if (cut > 0)
        hipLaunchKernelGGL(k_blk_minmax, dim3(std::min(nblocks(n), 2048)), dim3(TPB), 0, c->stream, c->d_x, c->d_y, n, cut, sc);
    LAUNCH(k_blk_keys, n, c->d_x, c->d_y, p, sc, c->keys_in.as<u64>(), c->vals_in.as<u32>());
    ev_record(c, 1);
    {
        size_t tmp_bytes = c->sort_tmp.bytes;
        const int begin_bit = 2 * p.rb;
        const int end_bit = begin_bit + p.nyb + std::max(1, bits_for((unsigned)p.R));
        hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tmp_bytes, c->keys_in.as<u64>(), c->keys_out.as<u64>(),
                                                 c->vals_in.as<u32>(), c->vals_out.as<u32>(), (size_t)n, begin_bit, end_bit, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs", hipGetErrorString(e));
    }
    u64* skeys = c->keys_out.as<u64>();
    u32* srow = c->vals_out.as<u32>();
    int* sx = (c->sv.as<int>() + SORT_PAD);
    int* sy = (c->sa.as<int>() + SORT_PAD);
    int* headflag = c->chainflag.as<int>();
    int* cidp1 = c->chainhead.as<int>();
    LAUNCH(k_blk_gather, n, p, skeys, sx, sy, headflag, sc);
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::inclusive_scan(c->scan_tmp.p, tb, headflag, cidp1, (size_t)n, rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan(cells)", hipGetErrorString(e));
    }
    int* cstart = c->b_cstart.as<int>();
    u64* ckey = c->b_ckey.as<u64>();
    int* cfirst = c->cellfirst.as<int>();
    int* rowcell = c->strip.as<int>();
    LAUNCH(k_blk_cells, n, p, skeys, headflag, cidp1, srow, sc, cstart, ckey, cfirst);
    LAUNCH(k_blk_rowtable, p.R + 1, p, sc, ckey, rowcell);
    ev_record(c, 2);
    int* nb = c->b_nb.as<int>();
    int* low = c->ncore.as<int>();
    int* alive = c->bsize.as<int>();
    double* cx = c->b_cx.as<double>();
    double* cy = c->b_cy.as<double>();
    int* linkbits = c->owner.as<int>();
    int* corec = c->state.as<int>();
    LAUNCH(k_blk_neighbors, n, p, sc, ckey, rowcell, cstart, sx, sy, nb, low, cx, cy);
    LAUNCH(k_blk_alive, n, sc, nb, low, cstart, alive, linkbits);
    hipLaunchKernelGGL(k_blk_links, dim3((unsigned)(((size_t)n * 4 + TPB - 1) / TPB)), dim3(TPB), 0, c->stream,
                       p, sc, nb, alive, cstart, sx, sy, cx, cy, linkbits);
    LAUNCH(k_blk_core, n, p, sc, nb, alive, cstart, linkbits, corec);
    ev_record(c, 3);
    LAUNCH(k_blk_union, n, sc, nb, linkbits, corec, c->parent.as<int>());
    hipLaunchKernelGGL(k_blk_flatten, dim3(nblocks(n, BIGTPB)), dim3(BIGTPB), 0, c->stream, sc, corec, c->parent.as<int>(), cfirst,
                       c->root.as<int>(), c->compkey.as<int>());
    ev_record(c, 4);
    LAUNCH(k_blk_rank_flags, n, sc, c->root.as<int>(), c->compkey.as<int>(), c->flag.as<int>());
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, c->flag.as<int>(), c->rankscan.as<int>(), 0, (size_t)n + 1,
                                               rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan", hipGetErrorString(e));
    }
    ev_record(c, 5);
    Table t = make_table(c);
    LAUNCH(k_init_table, n + 1, t, c->rankscan.as<int>(), n);
    int* clab = c->cnt.as<int>();
    LAUNCH(k_blk_cell_labels, n, sc, nb, linkbits, alive, c->root.as<int>(), c->compkey.as<int>(), c->rankscan.as<int>(), clab);
    hipLaunchKernelGGL(k_blk_point_labels, dim3(nblocks(n, BIGTPB)), dim3(BIGTPB), 0, c->stream, p, sc, cidp1, clab, srow, sx, sy, c->slot[c->cur].labels.as<int>(), t);
    HIP_TRY(hipGetLastError());
    // blockDBSCAN.py:74: an empty (fully filtered) mat raises only when the class is called
    // on it; pipe.py:64-65 returns before that, so cut > 0 with no survivors is just empty.
    { cl_chrom::Slot& sl = c->slot[c->cur]; sl.rows_valid = true; sl.sorted_src = false; sl.kmax = n; }      // ids <= cells <= n
    return finish_enqueue(c, p.R + 1, &sc->M, labels_out);
}

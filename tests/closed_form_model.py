"""Executable numpy/scipy model of the ORDER-FREE closed forms the HIP kernels implement
(rules R1-R3).  Test infrastructure: it lets the CPU-only test tier
check the *algorithm* the GPU runs (including the v2 release fix-up over the uncertain
set U and the v1 start-point rule) against the sequential oracle, without a GPU.
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree


def _pairs(X, Y, eps):
    """all unordered pairs with L1(orig) = Linf(rotated) <= eps (inclusive)."""
    R = np.stack([X - Y, X + Y], 1).astype(np.float64)
    if len(R) < 2:
        return np.zeros((0, 2), np.int64)
    return cKDTree(R).query_pairs(eps + 0.25, p=np.inf, output_type="ndarray").astype(np.int64)


def _core_components(n, pairs, minPts):
    deg = np.bincount(pairs.ravel(), minlength=n) + 1
    core = deg >= minPts
    cc = pairs[core[pairs[:, 0]] & core[pairs[:, 1]]]
    g = coo_matrix((np.ones(len(cc)), (cc[:, 0], cc[:, 1])), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    comp = np.where(core, comp, -1)
    return deg, core, comp


def _border_adjacency(pairs, core, comp):
    """-> (b, j) arrays: non-core point b adjacent to core point j."""
    a, b = pairs[:, 0], pairs[:, 1]
    m1 = core[a] & ~core[b]
    m2 = core[b] & ~core[a]
    bb = np.concatenate([b[m1], a[m2]])
    jj = np.concatenate([a[m1], b[m2]])
    return bb, jj


def labels_v2(X, Y, eps, minPts, stats=None):
    X = np.asarray(X, np.int64)
    Y = np.asarray(Y, np.int64)
    n = len(X)
    out = np.full(n, -1, np.int32)
    if n == 0:
        return out
    pairs = _pairs(X, Y, eps)
    deg, core, comp = _core_components(n, pairs, minPts)
    # rotated cells (cDBSCAN2.py:67-70); X <= Y so trunc == -floor(a/eps)
    a = Y - X
    v = X + Y
    assert (a >= 0).all() and (v >= 0).all()
    cellkey = (a // eps) * (1 << 40) + (v // eps)
    order = np.argsort(cellkey, kind="stable")
    uk, first_pos = np.unique(cellkey[order], return_index=True)
    cellfirst_of_key = order[first_pos]            # stable => first of run = min row
    cellfirst = cellfirst_of_key[np.searchsorted(uk, cellkey)]
    ncomp = int(comp.max()) + 1 if core.any() else 0
    if ncomp == 0:
        return out
    compkey = np.full(ncomp, n, np.int64)
    np.minimum.at(compkey, comp[core], cellfirst[core])
    ncore = np.bincount(comp[core], minlength=ncomp)
    bb, jj = _border_adjacency(pairs, core, comp)
    bc = comp[jj]
    # distinct (border, comp) pairs
    key = np.unique(bb * ncomp + bc)
    pb, pc = key // ncomp, key % ncomp
    # owner0 = min-key adjacent component
    owner0 = {}
    adj = {}
    for b_, c_ in zip(pb.tolist(), pc.tolist()):
        adj.setdefault(b_, []).append(c_)
    for b_, cs in adj.items():
        cs.sort(key=lambda c: compkey[c])
        owner0[b_] = cs[0]
    size0 = ncore.copy()
    for b_, c_ in owner0.items():
        size0[c_] += 1
    uncertain = (size0 < minPts) & (ncore > 0)
    dead = np.zeros(ncomp, bool)
    if uncertain.any():
        # sequential resolution over U in key order (the "release" rule, cDBSCAN2.py:180-183)
        recs = [cs for b_, cs in adj.items() if any(uncertain[c] for c in cs)]
        for c in sorted(np.where(uncertain)[0].tolist(), key=lambda c: compkey[c]):
            avail = 0
            for cs in recs:
                if c in cs:
                    lower = cs[: cs.index(c)]
                    if all(dead[x] for x in lower):
                        avail += 1
            if ncore[c] + avail < minPts:
                dead[c] = True
    if stats is not None:
        stats["uncertain"] = int(uncertain.sum())
        stats["released"] = int(dead.sum())
        stats["max_adj"] = max((len(cs) for cs in adj.values()), default=0)
    live = np.where(~dead & (ncore > 0))[0]
    rank = np.full(ncomp, -1, np.int64)
    rank[live[np.argsort(compkey[live])]] = np.arange(len(live))
    out[core] = rank[comp[core]]
    for b_, cs in adj.items():
        for c in cs:
            if not dead[c]:
                out[b_] = rank[c]
                break
    return out


def labels_v1(X, Y, eps, minPts, stats=None):
    X = np.asarray(X, np.int64)
    Y = np.asarray(Y, np.int64)
    n = len(X)
    out = np.full(n, -1, np.int32)
    pairs = _pairs(X, Y, eps)
    deg, core, comp = _core_components(n, pairs, minPts)
    if not core.any():
        return out
    ncomp = int(comp.max()) + 1
    compkey = np.full(ncomp, n, np.int64)           # min row of a core point = start point
    np.minimum.at(compkey, comp[core], np.where(core)[0])
    rank = np.empty(ncomp, np.int64)
    rank[np.argsort(compkey)] = np.arange(ncomp)
    out[core] = rank[comp[core]]
    bb, jj = _border_adjacency(pairs, core, comp)
    minS = {}
    maxT = {}
    for b_, j_ in zip(bb.tolist(), jj.tolist()):
        r = int(rank[comp[j_]])
        if b_ not in minS or r < minS[b_]:
            minS[b_] = r
        if compkey[comp[j_]] == j_:                 # j is its component's start point
            if b_ not in maxT or r > maxT[b_]:
                maxT[b_] = r
    steals = 0
    for b_, r in minS.items():
        if b_ in maxT:
            if maxT[b_] != r:
                steals += 1
            out[b_] = maxT[b_]
        else:
            out[b_] = r
    cnt = np.bincount(out[out >= 0], minlength=ncomp)
    small = cnt < minPts
    if stats is not None:
        stats["steals"] = steals
        stats["dropped"] = int((small & (cnt > 0)).sum())
    out[(out >= 0) & small[np.maximum(out, 0)]] = -1
    return out


def labels_block(X, Y, eps, minPts):
    X = np.asarray(X, np.int64)
    Y = np.asarray(Y, np.int64)
    n = len(X)
    out = np.full(n, -1, np.int32)
    nx = (X - X.min()) // eps + 1
    ny = (Y - Y.min()) // eps + 1
    key = nx * (1 << 40) + ny
    order = np.argsort(key, kind="stable")
    uk, first_pos, inv_sorted, counts = np.unique(key[order], return_index=True, return_inverse=True, return_counts=True)
    C = len(uk)
    cell_of = np.empty(n, np.int64)
    cell_of[order] = inv_sorted
    cfirst = order[first_pos]                         # min row of the cell = insertion order key
    cnx, cny = uk >> 40, uk & ((1 << 40) - 1)
    lut = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(cnx, cny))}
    nbrs = [[lut[(int(cnx[c]) + dx, int(cny[c]) + dy)] for dx in (-1, 0, 1) for dy in (-1, 0, 1)
             if (dx or dy) and (int(cnx[c]) + dx, int(cny[c]) + dy) in lut] for c in range(C)]
    tot = np.array([counts[c] + sum(counts[q] for q in nbrs[c]) for c in range(C)])
    low = tot < minPts
    alive = np.array([not (low[c] and all(low[q] for q in nbrs[c])) for c in range(C)])
    sx = np.bincount(cell_of, weights=None, minlength=C) * 0
    sumx = np.zeros(C, np.int64)
    sumy = np.zeros(C, np.int64)
    np.add.at(sumx, cell_of, X)
    np.add.at(sumy, cell_of, Y)
    cx = sumx.astype(np.float64) / counts.astype(np.float64)
    cy = sumy.astype(np.float64) / counts.astype(np.float64)
    members = [np.where(cell_of == c)[0] for c in range(C)] if C < 20000 else None
    if members is None:
        so = np.argsort(cell_of, kind="stable")
        bounds = np.searchsorted(cell_of[so], np.arange(C + 1))
        members = [so[bounds[c]:bounds[c + 1]] for c in range(C)]

    def linked(c, q):
        if abs(cx[c] - cx[q]) + abs(cy[c] - cy[q]) <= float(eps):
            return True
        pc, pq = members[c], members[q]
        d = np.abs(X[pc][:, None] - X[pq][None, :]) + np.abs(Y[pc][:, None] - Y[pq][None, :])
        return bool((d <= eps).any())

    links = [[] for _ in range(C)]
    for c in range(C):
        if not alive[c]:
            continue
        for q in nbrs[c]:
            if alive[q] and q > c and linked(c, q):
                links[c].append(q)
                links[q].append(c)
    psum = np.array([counts[c] + sum(counts[q] for q in links[c]) for c in range(C)])
    corec = alive & (psum >= minPts)
    rows, cols = [], []
    for c in range(C):
        if corec[c]:
            for q in links[c]:
                if corec[q]:
                    rows.append(c)
                    cols.append(q)
    g = coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(C, C))
    _, comp = connected_components(g, directed=False)
    comp = np.where(corec, comp, -1)
    if not corec.any():
        return out
    ncomp = comp.max() + 1
    compkey = np.full(ncomp, n, np.int64)
    np.minimum.at(compkey, comp[corec], cfirst[corec])
    present = np.unique(comp[corec])
    rank = np.full(ncomp, -1, np.int64)
    rank[present[np.argsort(compkey[present])]] = np.arange(len(present))
    clab = np.full(C, -1, np.int64)
    clab[corec] = rank[comp[corec]]
    for c in range(C):
        if alive[c] and not corec[c]:
            r = [rank[comp[q]] for q in links[c] if corec[q]]
            if r:
                clab[c] = max(r)
    out[:] = clab[cell_of]
    return out

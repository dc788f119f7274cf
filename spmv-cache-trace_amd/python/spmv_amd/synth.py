"""Synthetic sparse matrices of SURVEY.md 8(d), generated straight into CSR.

All generators return ``(rows, cols, row_ptr[int32], col_idx[int32], val[f64])``
with column indices ascending inside each row (what the reference's CSR
converter produces after its row-major sort, src/matrix/csr-matrix.cpp:200-237).
They are deterministic: same arguments, same arrays.
"""
import numpy as np


def poisson2d(n, row_begin=0, row_end=None):
    """5-point stencil on an n x n grid: N = n*n rows, Z = 5N - 4n entries.

    Row r = i*n + j has columns {r-n, r-1, r, r+1, r+n} where the neighbour
    exists on the grid, values {-1, -1, 4, -1, -1}.  n = 4096 is BASELINE.json's
    configs[1] (Z = 83 869 696).  With row_begin/row_end only that row range is
    generated (row_ptr rebased to 0, column indices still global): one rank's
    partition of the matrix.
    """
    N = n * n
    if row_end is None:
        row_end = N
    r = np.arange(row_begin, row_end, dtype=np.int32)
    nr = len(r)
    j = r % n
    i = r // n
    off = np.array([-n, -1, 0, 1, n], dtype=np.int32)
    cols = r[:, None] + off[None, :]
    mask = np.empty((nr, 5), dtype=bool)
    mask[:, 0] = i > 0
    mask[:, 1] = j > 0
    mask[:, 2] = True
    mask[:, 3] = j < n - 1
    mask[:, 4] = i < n - 1
    row_ptr = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(mask.sum(axis=1), out=row_ptr[1:])
    col_idx = cols[mask]
    vals = np.broadcast_to(np.array([-1.0, -1.0, 4.0, -1.0, -1.0]), (nr, 5))[mask]
    return nr, N, row_ptr.astype(np.int32), np.ascontiguousarray(col_idx, dtype=np.int32), \
        np.ascontiguousarray(vals, dtype=np.float64)


def banded(N, offsets, seed=1):
    """N x N matrix with one diagonal per entry of `offsets`, values U(-1,1)."""
    rng = np.random.default_rng(seed)
    off = np.array(sorted(set(int(o) for o in offsets)), dtype=np.int64)
    r = np.arange(N, dtype=np.int64)
    cols = r[:, None] + off[None, :]
    mask = (cols >= 0) & (cols < N)
    row_ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(mask.sum(axis=1), out=row_ptr[1:])
    col_idx = cols[mask].astype(np.int32)
    vals = rng.uniform(-1.0, 1.0, size=col_idx.shape[0])
    return N, N, row_ptr.astype(np.int32), col_idx, vals


def stencil27_like(nx, ny, nz, seed=2):
    """27-point stencil on an nx*ny*nz grid (nlpkkt-like: ~27 entries per row,
    three bands of bands).  Used as the stand-in for configs[3] (nlpkkt200),
    whose file cannot be fetched here."""
    N = nx * ny * nz
    offs = [dz * nx * ny + dy * nx + dx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    return banded(N, offs, seed=seed)


def random_uniform(N, M, k, seed=3):
    """k distinct uniformly random columns per row (sorted), values U(-1,1)."""
    rng = np.random.default_rng(seed)
    k = min(k, M)
    # sample with a random start + distinct strides to stay vectorised, then
    # fall back to per-row choice only for tiny M where collisions are likely
    if M >= 4 * k:
        cols = rng.integers(0, M, size=(N, k), dtype=np.int64)
        cols.sort(axis=1)
        # resolve duplicates by nudging (keeps sortedness, stays in range)
        for _ in range(8):
            dup = np.zeros_like(cols, dtype=bool)
            dup[:, 1:] = cols[:, 1:] <= cols[:, :-1]
            if not dup.any():
                break
            cols[dup] = cols[dup] + 1
            cols = np.minimum(cols, M - 1)
            cols.sort(axis=1)
        keep = np.ones_like(cols, dtype=bool)
        keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    else:
        cols = np.stack([np.sort(rng.choice(M, size=k, replace=False)) for _ in range(N)])
        keep = np.ones_like(cols, dtype=bool)
    row_ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(keep.sum(axis=1), out=row_ptr[1:])
    col_idx = cols[keep].astype(np.int32)
    vals = rng.uniform(-1.0, 1.0, size=col_idx.shape[0])
    return N, M, row_ptr.astype(np.int32), col_idx, vals


def powerlaw(N, M, alpha=1.8, max_len=4700, seed=4, mean_target=3.1):
    """webbase-like: row lengths ~ Zipf(alpha) capped at max_len, random columns.
    Includes empty rows and a few very long rows."""
    rng = np.random.default_rng(seed)
    lens = rng.zipf(alpha, size=N).astype(np.int64)
    lens = np.minimum(lens, min(max_len, M))
    # thin to roughly the requested mean, keep some empty rows
    scale = mean_target / max(lens.mean(), 1e-9)
    if scale < 1.0:
        lens = np.floor(lens * scale + rng.uniform(0, 1, size=N)).astype(np.int64)
    lens[rng.integers(0, N, size=max(1, N // 50))] = 0
    if N > 10:
        lens[rng.integers(0, N, size=3)] = min(max_len, M)
    row_ptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(lens, out=row_ptr[1:])
    Z = int(row_ptr[-1])
    # random columns, sorted inside each row; duplicate (row, col) pairs may
    # occur and are kept (the reference keeps duplicates too)
    rows_of = np.repeat(np.arange(N, dtype=np.int64), lens)
    rc = rng.integers(0, M, size=Z, dtype=np.int64)
    order = np.lexsort((rc, rows_of))
    col_idx = rc[order].astype(np.int32)
    vals = rng.uniform(-1.0, 1.0, size=Z)
    return N, M, row_ptr.astype(np.int32), col_idx, vals


def x_vector(M, kind="uniform", seed=12345):
    """`ones` is what the reference CLI multiplies by (src/kernels/csr-spmv.cpp:35);
    `uniform` is the meaningful parity input."""
    if kind == "ones":
        return np.ones(M)
    return np.random.default_rng(seed).uniform(-1.0, 1.0, size=M)


def csr_to_coordinate(rows, row_ptr, col_idx, val):
    """1-based coordinate triplets in row-major order."""
    lens = np.diff(row_ptr.astype(np.int64))
    i = np.repeat(np.arange(1, rows + 1, dtype=np.int32), lens)
    return i, (col_idx + 1).astype(np.int32), val


def write_mtx(path, rows, cols, i, j, a, field="real", symmetry="general", comments=()):
    """Write coordinate entries as a Matrix Market file (17 significant digits)."""
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate %s %s\n" % (field, symmetry))
        for c in comments:
            f.write("%" + c + "\n")
        f.write("%d %d %d\n" % (rows, cols, len(i)))
        if field == "pattern":
            for r, c in zip(i, j):
                f.write("%d %d\n" % (r, c))
        elif field == "integer":
            for r, c, v in zip(i, j, a):
                f.write("%d %d %d\n" % (r, c, int(v)))
        else:
            for r, c, v in zip(i, j, a):
                f.write("%d %d %.17g\n" % (r, c, v))


def csr_bytes(rows, cols, nnz):
    """Algorithmic bytes of one CSR y += A*x (SURVEY 8d / BASELINE.md 3)."""
    return 12 * nnz + 4 * (rows + 1) + 16 * rows + 8 * cols


def coo_bytes(rows, cols, nnz):
    return 16 * nnz + 16 * rows + 8 * cols


def ell_bytes(rows, cols, row_length):
    return 12 * rows * row_length + 16 * rows + 8 * cols


def mesh_dofs(grid, d, seed=1, jitter_share=0.3):
    """An unstructured-looking mesh: nodes of a gx x gy x gz grid, each linked to its 27 grid neighbours, 30 % of the links ending up to
    3 nodes beside their grid neighbour (no two rows are shifted copies of each other); d unknowns per node, numbered node by node:
    dense d x d blocks, the d rows of a node with the same columns.  (d = 3: what synthetic:queen is; d = 2 / 4: no 3 x 3 blocks.)"""
    rng = np.random.default_rng(seed)
    gx, gy, gz = grid
    n = gx * gy * gz
    z, y, x = np.meshgrid(np.arange(gz), np.arange(gy), np.arange(gx), indexing="ij")
    z, y, x = z.ravel(), y.ravel(), x.ravel()
    rows, cols = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                zz, yy, xx = z + dz, y + dy, x + dx
                good = (zz >= 0) & (zz < gz) & (yy >= 0) & (yy < gy) & (xx >= 0) & (xx < gx)
                c = (zz * gy + yy) * gx + xx
                if (dz, dy, dx) != (0, 0, 0):
                    moved = rng.random(n) < jitter_share
                    c = np.where(moved, c + rng.integers(-3, 4, size=n), c)
                    good &= (c >= 0) & (c < n)
                rows.append(np.arange(n)[good])
                cols.append(c[good])
    key = np.unique(np.concatenate(rows).astype(np.int64) * n + np.concatenate(cols))
    rn, cn = key // n, key % n
    lens_n = np.bincount(rn, minlength=n)
    pn = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens_n, out=pn[1:])
    seg = ((cn * d)[:, None] + np.arange(d)).ravel()          # node i: its row's columns, pn[i] * d ... pn[i + 1] * d
    node = np.repeat(np.arange(n), d)                          # expanded row -> node
    starts, lens = pn[node] * d, lens_n[node] * d
    p = np.zeros(n * d + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    idx = np.repeat(starts - p[:-1], lens) + np.arange(int(p[-1]))
    c = seg[idx].astype(np.int32)
    return n * d, n * d, p.astype(np.int32), c, rng.uniform(-1, 1, size=len(c))


def delaunay_mesh(npoints, d=3, seed=1, order="rcm", dim=3):
    """An UNSTRUCTURED finite-element matrix with variable valence (round 6; VERDICT r05 item 4: a structure no generator of the
    host library shaped): the Delaunay tetrahedra (dim = 3; triangles for dim = 2) of `npoints` uniformly random points of the unit
    cube, one node per point, two nodes coupled where they share an element, `d` unknowns per node (dense d x d blocks, as a
    vector-valued P1 discretisation assembles them), numbered by reverse Cuthill-McKee (order = "rcm", scipy's) -- or left in the
    random order of the points (order = "random": the scattered class) -- unknown by unknown within a node.  A 3-D Delaunay node has
    ~15.5 neighbours on average, between 4 and ~40: rows of 16.5 d entries on average, no two neighbourhoods alike, nothing
    shifted, nothing periodic.  Values U(-1, 1) with a dominant diagonal; the structure is symmetric.  Returns CSR arrays with
    ascending columns.  Needs scipy (test / tool infrastructure only)."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    pts = rng.random((npoints, dim))
    tri = Delaunay(pts)
    simp = tri.simplices.astype(np.int64)
    k = simp.shape[1]
    a = np.concatenate([simp[:, i] for i in range(k) for j in range(k) if i != j])
    b = np.concatenate([simp[:, j] for i in range(k) for j in range(k) if i != j])
    del tri, simp
    G = sp.coo_matrix((np.ones(len(a), dtype=np.int8), (a, b)), shape=(npoints, npoints)).tocsr()
    del a, b
    G.data[:] = 1  # (duplicates were summed)
    G = (G + sp.identity(npoints, dtype=np.int8, format="csr")).tocsr()
    if order == "rcm":
        perm = reverse_cuthill_mckee(G, symmetric_mode=True)
        G = G[perm][:, perm].tocsr()
    G.sort_indices()
    P, J = G.indptr.astype(np.int64), G.indices.astype(np.int64)
    deg = np.diff(P)
    if d == 1:
        p, c = P, J
    else:
        # every node's column list, d columns per neighbour, repeated for each of its d rows
        node_cols = (J[:, None] * d + np.arange(d)[None, :]).reshape(-1)          # node-major: d * deg(i) columns per node
        starts = np.repeat(P[:-1] * d, d)                                           # where a row's node's list starts ...
        lens = np.repeat(deg * d, d)                                                # ... and how long it is, for every row
        p = np.zeros(npoints * d + 1, dtype=np.int64)
        np.cumsum(lens, out=p[1:])
        within = np.arange(int(p[-1]), dtype=np.int64) - np.repeat(p[:-1], lens)
        c = node_cols[np.repeat(starts, lens) + within]
    if int(p[-1]) > 2 ** 31 - 1:
        raise ValueError("delaunay_mesh: more than 2^31 - 1 entries")
    rows = npoints * d
    v = rng.uniform(-1.0, 1.0, size=int(p[-1]))
    r = np.repeat(np.arange(rows, dtype=np.int64), np.diff(p))
    v[c == r] += 4.0 * d
    return rows, rows, p.astype(np.int32), c.astype(np.int32), v

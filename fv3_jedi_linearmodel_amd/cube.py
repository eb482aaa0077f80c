"""Cubed-sphere topology, halo-exchange index tables and gnomonic metric terms (host side).

The reference gets all of this from outside the hot path: the 12 face contacts are handed to FMS
(`mpp_define_mosaic`, TOOLS/fv_mp_nlm_mod.F90:524-572), the data motion of
`mpp_update_domains` lives inside FMS, and the metric terms come from `grid_utils_init`
(NLM/fv_grid_utils_nlm.F90:78-840).  None of it is buildable here, so this module derives the same
objects from first principles — six face frames on the unit cube that reproduce the contact table
exactly (checked by `check_contacts`), halo points found by folding the neighbour face around the
shared edge, vector components re-expressed in the neighbour's basis (that is where the D-/C-grid
component swap and sign flip across rotated contacts come from) — and is shared by the oracle and
the product (it is data, not arithmetic).  "Parity unpinned" for the exchange:
covered by invariants in tests/test_cube.py (analytic scalar continuity, solid-body wind continuity,
exchange/adjoint-exchange dot product).
"""
import numpy as np

NG = 3
X, Y, Z = np.eye(3)
# tile -> (normal, e1 (i direction), e2 (j direction)); e1 x e2 = normal
FRAMES = {1: (X, Y, Z), 2: (Y, -X, Z), 3: (Z, -X, -Y), 4: (-X, -Z, -Y), 5: (-Y, -Z, X), 6: (-Z, Y, X)}
# (tile1, edge1, tile2, edge2, reversed)
CONTACTS = [(1, "E", 2, "W", 0), (1, "N", 3, "W", 1), (1, "W", 5, "N", 1), (1, "S", 6, "N", 0), (2, "N", 3, "S", 0),
            (2, "E", 4, "S", 1), (2, "S", 6, "E", 1), (3, "E", 4, "W", 0), (3, "N", 5, "W", 1), (4, "N", 5, "S", 0),
            (4, "E", 6, "S", 1), (5, "E", 6, "W", 0)]
# location kinds in index space: offset of element (i,j) from (i-1, j-1)
LOC = {"center": (0.5, 0.5), "corner": (0.0, 0.0), "yedge": (0.5, 0.0), "xedge": (0.0, 0.5)}


def _fold(tile, x, y, n, corners=False):
    """Index-space point (x,y) of `tile` (tile covers [0,n]^2) -> (tile', x', y', R) where R maps the
    local direction components (d1,d2) of `tile` at that point to those of tile' (2x2 with entries 0,+-1).
    Points inside the tile map to themselves.  Corner regions (both coordinates outside) have no
    neighbour: None, unless corners=True, which gives the "x-direction view" used for geometry only —
    the row is continued across the south/north neighbour into the third face at that vertex (the
    cells copy_corners(dir=1) takes its values from, tp_core_tlm.F90:2046-2079)."""
    nrm, e1, e2 = FRAMES[tile]
    ox, oy = (x < 0) or (x > n), (y < 0) or (y > n)
    if not ox and not oy:
        return tile, x, y, np.eye(2)
    if ox and oy and not corners:
        return None
    a, b = 2.0 * x / n - 1.0, 2.0 * y / n - 1.0
    if oy and y > n:
        P = nrm + a * e1 + e2 - (b - 1.0) * nrm; d1, d2, nb = e1, -nrm, e2
    elif oy:
        P = nrm + a * e1 - e2 - (-1.0 - b) * nrm; d1, d2, nb = e1, nrm, -e2
    elif x > n:
        P = nrm + e1 + b * e2 - (a - 1.0) * nrm; d1, d2, nb = -nrm, e2, e1
    else:
        P = nrm - e1 + b * e2 - (-1.0 - a) * nrm; d1, d2, nb = nrm, e2, -e1
    for t2, (n2, f1, f2) in FRAMES.items():
        if np.allclose(n2, nb):      # the neighbour across that edge (points on an edge line belong to it too)
            xb, yb = ((P - n2) @ f1 + 1.0) * n / 2.0, ((P - n2) @ f2 + 1.0) * n / 2.0
            R = np.rint(np.array([[d1 @ f1, d2 @ f1], [d1 @ f2, d2 @ f2]]))
            if ox and oy:            # still outside the neighbour along the row: continue into the third face
                t3, x3, y3, R2 = _fold(t2, xb, yb, n)
                return t3, x3, y3, R2 @ R
            return t2, xb, yb, R
    raise RuntimeError("fold failed")


def check_contacts(n=8):
    """The frames reproduce SURVEY.md A.3 (edge pairing and index orientation of all 12 contacts)."""
    for t1, e1, t2, e2, rev in CONTACTS:
        for a in (1, n):
            x, y = {"E": (n + 0.5, a - 0.5), "W": (-0.5, a - 0.5), "N": (a - 0.5, n + 0.5), "S": (a - 0.5, -0.5)}[e1]
            tb, xb, yb, _ = _fold(t1, x, y, n)
            assert tb == t2, (t1, e1, tb, t2)
            ab = a if not rev else n + 1 - a
            exp = {"E": (n - 0.5, ab - 0.5), "W": (0.5, ab - 0.5), "N": (ab - 0.5, n - 0.5), "S": (ab - 0.5, 0.5)}[e2]
            assert abs(xb - exp[0]) < 1e-9 and abs(yb - exp[1]) < 1e-9, (t1, e1, t2, e2, rev, xb, yb, exp)
    return True


def plane_index(i, j, n):
    """flat index in the padded plane (isd:ied+1, jsd:jed+1) of an n x n tile"""
    return (j + NG - 1) * (n + 2 * NG + 1) + (i + NG - 1)


def exchange_table(n, fields):
    """Halo-exchange gather table for a group of co-located fields.

    fields: list of (location kind, direction) with direction in {"s" scalar, "1" e1 component,
    "2" e2 component}, e.g. D-grid winds [("yedge","1"), ("xedge","2")], C-grid winds
    [("xedge","1"), ("yedge","2")], a cell scalar [("center","s")], divg_d [("corner","s")].
    Returns int32 array of rows (dst_field, dst_tile0, dst_index, src_field, src_tile0, src_index, sign):
    every halo element (<= NG cells outside, corner regions excluded) of every tile, each taking
    sign * the value of exactly one element of a neighbouring tile."""
    rows = []
    lo, hi = 1 - NG, n + NG
    for tile in range(1, 7):
        for fi, (kind, dirn) in enumerate(fields):
            ox, oy = LOC[kind]
            imax = hi + (1 if ox == 0.0 else 0); jmax = hi + (1 if oy == 0.0 else 0)
            for j in range(lo, jmax + 1):
                for i in range(lo, imax + 1):
                    x, y = i - 1 + ox, j - 1 + oy
                    if 0 <= x <= n and 0 <= y <= n:
                        continue
                    r = _fold(tile, x, y, n)
                    if r is None:
                        continue
                    tb, xb, yb, R = r
                    if dirn == "s":
                        sf, sign = fi, 1
                        kb = kind if R[0, 0] != 0 else {"yedge": "xedge", "xedge": "yedge"}.get(kind, kind)
                    else:
                        comp = 0 if dirn == "1" else 1          # which local component this field is
                        col = R[:, comp]                          # that direction in the neighbour's basis
                        cb = int(np.argmax(np.abs(col))); sign = int(col[cb])
                        kb = kind if R[0, 0] != 0 else {"yedge": "xedge", "xedge": "yedge"}[kind]
                        want = (kb, "1" if cb == 0 else "2")
                        sf = fields.index(want)
                    oxb, oyb = LOC[kb]
                    ib, jb = int(round(xb - oxb)) + 1, int(round(yb - oyb)) + 1
                    assert abs(ib - 1 + oxb - xb) < 1e-9 and abs(jb - 1 + oyb - yb) < 1e-9
                    rows.append((fi, tile - 1, plane_index(i, j, n), sf, tb - 1, plane_index(ib, jb, n), sign))
    return np.array(rows, dtype=np.int32)


def boundary_table(n):
    """mpp_get_boundary of the D-grid winds (dyn_core_tlm.F90:2418-2431): the shared edge rows u(1:n, npy) and
    v(npx, 1:n) of every face take the neighbour's value of the same physical edge (its u(:,1) or v(1,:), swapped and
    sign-changed where the contact is rotated).  Same row format as exchange_table, fields [u, v]."""
    fields = [("yedge", "1"), ("xedge", "2")]
    rows = []
    eps = 1.0e-6
    for tile in range(1, 7):
        for fi, pts in ((0, [(i, n + 1, i - 0.5, n + eps) for i in range(1, n + 1)]), (1, [(n + 1, j, n + eps, j - 0.5) for j in range(1, n + 1)])):
            kind, dirn = fields[fi]
            for (i, j, x, y) in pts:
                tb, xb, yb, R = _fold(tile, x, y, n)
                comp = 0 if dirn == "1" else 1
                col = R[:, comp]
                cb = int(np.argmax(np.abs(col))); sign = int(col[cb])
                kb = kind if R[0, 0] != 0 else {"yedge": "xedge", "xedge": "yedge"}[kind]
                sf = fields.index((kb, "1" if cb == 0 else "2"))
                oxb, oyb = LOC[kb]
                ib, jb = int(round(xb - oxb)) + 1, int(round(yb - oyb)) + 1
                assert abs(ib - 1 + oxb - xb) < 1e-4 and abs(jb - 1 + oyb - yb) < 1e-4
                assert (ib == 1 and kb == "xedge") or (jb == 1 and kb == "yedge"), (tile, i, j, tb, ib, jb, kb)
                rows.append((fi, tile - 1, plane_index(i, j, n), sf, tb - 1, plane_index(ib, jb, n), sign))
    return np.array(rows, dtype=np.int32)


EXCHANGE_FIELDS = {"cell": [("center", "s")], "dvec": [("yedge", "1"), ("xedge", "2")], "cvec": [("xedge", "1"), ("yedge", "2")],
                   "corner": [("corner", "s")]}


def all_tables(n):
    """the five exchange tables of include/fv3lm.h fv3lm_set_exchange, keyed by kind name"""
    t = {k: exchange_table(n, f) for k, f in EXCHANGE_FIELDS.items()}
    t["dedge"] = boundary_table(n)
    return t


def apply_table(tab, f0, f1=None):
    """numpy reference of one exchange: fields [6, nk, pj, pi] updated in place"""
    fl = [f0.reshape(6, f0.shape[1], -1)] + ([f1.reshape(6, f1.shape[1], -1)] if f1 is not None else [])
    vals = [None] * len(tab)
    src = [fl[r[3]][r[4], :, r[5]] * r[6] for r in tab]
    for r, v in zip(tab, src):
        fl[r[0]][r[1], :, r[2]] = v


def adjoint_table(tab):
    """CSR form of the transposed exchange: for every source element the halo elements that copy it.
    Returns (src_keys[ns,3] = field, tile, index; ptr[ns+1]; dst[nnz,4] = field, tile, index, sign)."""
    key = (tab[:, 3].astype(np.int64) << 40) | (tab[:, 4].astype(np.int64) << 32) | tab[:, 5].astype(np.int64)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    uniq, start = np.unique(ks, return_index=True)
    ptr = np.append(start, len(ks)).astype(np.int32)
    src = np.stack([(uniq >> 40) & 0xFF, (uniq >> 32) & 0xFF, uniq & 0xFFFFFFFF], axis=1).astype(np.int32)
    dst = tab[order][:, [0, 1, 2, 6]].astype(np.int32)
    return src, ptr, np.ascontiguousarray(dst)


# --------------------------------------------------------------------------------- sub-face tiles (layout > 1 x 1)
# fv_flags_type%layout (NLM/fv_control_nlm.F90:556) cuts every face into layout x layout tiles; rank -> tile map as in
# tools/fv_mp_nlm_mod.F90:452-453 (face-major, then rows of tiles).  A tile is the window is..ie x js..je of its face in the face's global
# indices; its padded plane is indexed like a face's, relative to (is, js).
def tiles(n, layout, order="position"):
    """[(face0, is, js)] of the 6 * layout^2 tiles of a C<n> cube, nt = n // layout cells each.  order "face": the reference's rank order
    (face-major, tools/fv_mp_nlm_mod.F90:452-453: what a host with one tile per MPI rank hands over); "position" (default of the harness):
    the six tiles at the same place of their faces next to each other -- tiles with the same window form one class of launches in the
    library (dycore.h set_class), so a GPU holding several tiles runs each stage once per position, not once per tile."""
    assert layout >= 1 and n % layout == 0, (n, layout)
    nt = n // layout
    if order == "face":
        return [(f, 1 + bi * nt, 1 + bj * nt) for f in range(6) for bj in range(layout) for bi in range(layout)]
    return [(f, 1 + bi * nt, 1 + bj * nt) for bj in range(layout) for bi in range(layout) for f in range(6)]


def tile_window(a, tl, nt):
    """the padded-plane windows [ntile, ..., nt+7, nt+7] of face arrays a[6, ..., n+7, n+7] (halo included: global smooth fields)"""
    return np.ascontiguousarray(np.stack([a[f][..., j0 - 1:j0 - 1 + nt + 2 * NG + 1, i0 - 1:i0 - 1 + nt + 2 * NG + 1] for (f, i0, j0) in tl]))


def tile_gather(w, tl, n, nt, lead=()):
    """inverse of tile_window on the compute domains: face arrays [6, ..., n+7, n+7] with every cell from the tile that owns it (rest zero)"""
    out = np.zeros((6,) + tuple(w.shape[1:-2]) + (n + 2 * NG + 1, n + 2 * NG + 1))
    for t, (f, i0, j0) in enumerate(tl):
        out[f][..., j0 + NG - 1:j0 + NG - 1 + nt, i0 + NG - 1:i0 + NG - 1 + nt] = w[t][..., NG:NG + nt, NG:NG + nt]
    return out


_STAG = {"center": (0, 0), "corner": (1, 1), "yedge": (0, 1), "xedge": (1, 0)}      # extra points of the compute domain in i, j


def tiled_tables(n, layout):
    """The five exchange tables for the 6 * layout^2 tiles.  A point of a tile's padded plane outside the tile's own compute domain
    takes the value of the tile that owns it: the same face point where it lies inside the face, the face-level table's source
    (the neighbour face, component swap and sign included) where it lies beyond a cube edge; the corner squares beyond a cube
    corner hold no data.  'dedge': the D-grid edge rows u(is:ie, je+1), v(ie+1, js:je) from the tile whose low edge they are."""
    nt = n // layout
    tl = tiles(n, layout)
    pn, pt = n + 2 * NG + 1, nt + 2 * NG + 1
    face_tabs = all_tables(n)
    tidx = {(f, i0, j0): t for t, (f, i0, j0) in enumerate(tl)}

    def owner(f, kind, i, j):       # tile that holds face point (i, j) of a field at `kind` in its compute domain (low edge of shared points)
        bi = min((i - 1) // nt, layout - 1); bj = min((j - 1) // nt, layout - 1)
        return tidx[(f, 1 + bi * nt, 1 + bj * nt)]

    def pidx(t, i, j):
        _, i0, j0 = tl[t]
        return (j - j0 + NG) * pt + (i - i0 + NG)
    out = {}
    for kname, fields in EXCHANGE_FIELDS.items():
        canon = {}
        for r in face_tabs[kname]:
            canon[(int(r[0]), int(r[1]), int(r[2]))] = (int(r[3]), int(r[4]), int(r[5]), int(r[6]))
        rows = []
        for t, (f, i0, j0) in enumerate(tl):
            for fi, (kind, _) in enumerate(fields):
                ei, ej = _STAG[kind]
                for j in range(j0 - NG, j0 + nt + NG + 1):
                    for i in range(i0 - NG, i0 + nt + NG + 1):
                        if i0 <= i <= i0 + nt - 1 + ei and j0 <= j <= j0 + nt - 1 + ej:
                            continue                                  # own compute domain
                        if i < 1 - NG or i > n + NG + 1 or j < 1 - NG or j > n + NG + 1:
                            continue
                        if 1 <= i <= n + ei and 1 <= j <= n + ej:     # inside the face: the owning tile's copy
                            sf, sface, si, sj, sg = fi, f, i, j, 1
                        else:
                            c = canon.get((fi, f, (j + NG - 1) * pn + (i + NG - 1)))
                            if c is None:
                                continue                              # beyond a cube corner, or not part of this exchange
                            sf, sface, sidx, sg = c
                            sj, si = divmod(sidx, pn); si += 1 - NG; sj += 1 - NG
                        ts = owner(sface, fields[sf][0], si, sj)
                        rows.append((fi, t, pidx(t, i, j), sf, ts, pidx(ts, si, sj), sg))
        out[kname] = np.array(rows, dtype=np.int32).reshape(-1, 7)
    # shared edge rows
    canon = {(int(r[0]), int(r[1]), int(r[2])): (int(r[3]), int(r[4]), int(r[5]), int(r[6])) for r in face_tabs["dedge"]}
    fields = EXCHANGE_FIELDS["dvec"]
    rows = []
    for t, (f, i0, j0) in enumerate(tl):
        for fi, pts in ((0, [(i, j0 + nt) for i in range(i0, i0 + nt)]), (1, [(i0 + nt, j) for j in range(j0, j0 + nt)])):
            for (i, j) in pts:
                if (fi == 0 and j <= n) or (fi == 1 and i <= n):      # interior tile boundary: the neighbour's low edge, same face point
                    sf, sface, si, sj, sg = fi, f, i, j, 1
                else:
                    sf, sface, sidx, sg = canon[(fi, f, (j + NG - 1) * pn + (i + NG - 1))]
                    sj, si = divmod(sidx, pn); si += 1 - NG; sj += 1 - NG
                ts = owner(sface, fields[sf][0], si, sj)
                rows.append((fi, t, pidx(t, i, j), sf, ts, pidx(ts, si, sj), sg))
    out["dedge"] = np.array(rows, dtype=np.int32).reshape(-1, 7)
    return out


# --------------------------------------------------------------------------------------------- grid
def _norm(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _dist(a, b):
    """great-circle angle between unit vectors (chord formula, accurate for small angles)"""
    return 2.0 * np.arcsin(np.clip(0.5 * np.linalg.norm(a - b, axis=-1), 0.0, 1.0))


def _tri_area(a, b, c):
    num = np.abs(np.einsum("...i,...i", a, np.cross(b, c)))
    den = 1.0 + np.einsum("...i,...i", a, b) + np.einsum("...i,...i", b, c) + np.einsum("...i,...i", c, a)
    return 2.0 * np.arctan2(num, den)


def _quad_area(p00, p10, p11, p01):
    return _tri_area(p00, p10, p11) + _tri_area(p00, p11, p01)


def _tangent(at, frm, to):
    """unit tangent at `at` pointing from `frm` towards `to`"""
    d = to - frm
    d = d - np.einsum("...i,...i", d, at)[..., None] * at
    return _norm(d)


def cube_corner_points(n):
    """Unit vectors of the cell corners on the padded plane [6, pj, pi, 3]: equiangular gnomonic faces,
    halo corners taken from the neighbouring faces, corner regions by planar extrapolation."""
    pj = n + 2 * NG + 1
    ang = -np.pi / 4 + (np.pi / 2) * np.arange(n + 1) / n
    t = np.tan(ang)
    P = np.full((6, pj, pj, 3), np.nan)
    for tile, (nrm, e1, e2) in FRAMES.items():
        pts = nrm[None, None, :] + t[None, :, None] * e1[None, None, :] + t[:, None, None] * e2[None, None, :]
        P[tile - 1, NG:NG + n + 1, NG:NG + n + 1] = _norm(pts)
    tab = exchange_table(n, [("corner", "s")])
    flat = P.reshape(6, pj * pj, 3)
    flat[tab[:, 1], tab[:, 2]] = flat[tab[:, 4], tab[:, 5]]
    # corner regions: x-direction view (see _fold) — real points of the third face at that vertex
    for tile in range(1, 7):
        for j in list(range(1 - NG, 1)) + list(range(n + 2, n + NG + 2)):
            for i in list(range(1 - NG, 1)) + list(range(n + 2, n + NG + 2)):
                t3, x3, y3, _ = _fold(tile, i - 1.0, j - 1.0, n, corners=True)
                P[tile - 1, j + NG - 1, i + NG - 1] = P[t3 - 1, int(round(y3)) + NG, int(round(x3)) + NG]
    return P


def cubed_sphere_metrics(n, radius=6371.0e3, omega=7.292e-5):
    """All metric planes of include/fv3lm.h for the six faces, [6, pj, pi] each, plus da_min, da_min_c,
    the a2b edge weights edge_w/e/s/n [6, pj] and the extrap_corner coefficients [6, 4, 3]
    (definitions: NLM/fv_grid_utils_nlm.F90:455-745, a2b_edge_tlm.F90:1478-1487)."""
    from .grid import METRIC_NAMES
    pj = n + 2 * NG + 1
    P = cube_corner_points(n)                                   # corners (i,j) -> index [j+2, i+2]
    m = {k: np.zeros((6, pj, pj)) for k in METRIC_NAMES}
    c00, c10, c01, c11 = P[:, :-1, :-1], P[:, :-1, 1:], P[:, 1:, :-1], P[:, 1:, 1:]
    ctr = _norm(c00 + c10 + c01 + c11)                          # cell centres [6, pj-1, pj-1]
    mw, me, ms, mn = _norm(c00 + c01), _norm(c10 + c11), _norm(c00 + c10), _norm(c01 + c11)
    sl = (slice(None), slice(0, pj - 1), slice(0, pj - 1))
    m["dx"][:, :, :-1] = radius * _dist(P[:, :, :-1], P[:, :, 1:])      # south edge of cell (i,j), rows j up to jed+1
    m["dy"][:, :-1, :] = radius * _dist(P[:, :-1, :], P[:, 1:, :])
    m["dxa"][sl] = radius * _dist(mw, me); m["dya"][sl] = radius * _dist(ms, mn)
    m["dxc"][:, :-1, 1:-1] = radius * _dist(ctr[:, :, :-1], ctr[:, :, 1:])
    m["dyc"][:, 1:-1, :-1] = radius * _dist(ctr[:, :-1, :], ctr[:, 1:, :])
    m["area"][sl] = radius ** 2 * _quad_area(c00, c10, c11, c01)
    area_c = np.zeros((6, pj, pj))
    area_c[:, 1:-1, 1:-1] = radius ** 2 * _quad_area(ctr[:, :-1, :-1], ctr[:, :-1, 1:], ctr[:, 1:, 1:], ctr[:, 1:, :-1])
    # angles between the coordinate lines seen from inside cell (i,j): 1 west, 2 south, 3 east, 4 north edge
    # mid-points, 5 centre, 6..9 corners sw, se, ne, nw
    def ang(at, e1, e2):
        c = np.einsum("...i,...i", e1, e2)
        return np.sqrt(np.maximum(0.0, 1.0 - c * c)), c
    views = {1: (mw, _tangent(mw, mw, ctr), _tangent(mw, c00, c01)), 3: (me, _tangent(me, ctr, me), _tangent(me, c10, c11)),
             2: (ms, _tangent(ms, c00, c10), _tangent(ms, ms, ctr)), 4: (mn, _tangent(mn, c01, c11), _tangent(mn, ctr, mn)),
             5: (ctr, _tangent(ctr, mw, me), _tangent(ctr, ms, mn)),
             6: (c00, _tangent(c00, c00, c10), _tangent(c00, c00, c01)), 7: (c10, _tangent(c10, c00, c10), _tangent(c10, c10, c11)),
             8: (c11, _tangent(c11, c01, c11), _tangent(c11, c10, c11)), 9: (c01, _tangent(c01, c01, c11), _tangent(c01, c00, c01))}
    for k, (at, e1, e2) in views.items():
        s, c = ang(at, e1, e2)
        m["sin_sg%d" % k][sl] = s; m["cos_sg%d" % k][sl] = c
    # The cell diagonal to a cube vertex (0,0), (npx,0), (0,npy), (npx,npy) does not exist on the sphere (two
    # of its corners coincide): give it the geometry of the cell copy_corners(dir=1) would read there.
    npx_ = n + 1
    for (di, dj, si, sj) in ((0, 0, 0, 1), (npx_, 0, npx_, 1), (npx_, npx_, npx_, npx_ - 1), (0, npx_, 0, npx_ - 1)):
        for nm in ["area"] + ["sin_sg%d" % k for k in range(1, 10)] + ["cos_sg%d" % k for k in range(1, 10)]:
            m[nm][:, dj + NG - 1, di + NG - 1] = m[nm][:, sj + NG - 1, si + NG - 1]
    for k in range(1, 10):                                      # unused last row/column: benign values
        for nm, fillv in (("sin_sg%d" % k, 1.0), ("cos_sg%d" % k, 0.0)):
            m[nm][:, -1, :] = fillv; m[nm][:, :, -1] = fillv
    for nm in ("dx", "dy", "dxa", "dya", "dxc", "dyc", "area"):
        a = m[nm]; a[a == 0.0] = a[a > 0].mean()
    area_c[area_c == 0.0] = area_c[area_c > 0].mean()
    for nm in ("dx", "dy", "dxa", "dya", "dxc", "dyc"):
        m["r" + nm] = 1.0 / m[nm]
    m["rarea"] = 1.0 / m["area"]; m["rarea_c"] = 1.0 / area_c
    s, c = m, None
    # cosa_u/sina_u (west edge of cell i), cosa_v/sina_v (south edge), cosa/sina (corner): :490-517
    m["cosa_u"][:, :, 1:] = 0.5 * (m["cos_sg3"][:, :, :-1] + m["cos_sg1"][:, :, 1:])
    m["sina_u"][:, :, 1:] = 0.5 * (m["sin_sg3"][:, :, :-1] + m["sin_sg1"][:, :, 1:])
    m["cosa_v"][:, 1:, :] = 0.5 * (m["cos_sg4"][:, :-1, :] + m["cos_sg2"][:, 1:, :])
    m["sina_v"][:, 1:, :] = 0.5 * (m["sin_sg4"][:, :-1, :] + m["sin_sg2"][:, 1:, :])
    m["sina_u"][:, :, 0] = 1.0; m["sina_v"][:, 0, :] = 1.0
    m["cosa"][:, 1:, 1:] = 0.5 * (m["cos_sg8"][:, :-1, :-1] + m["cos_sg6"][:, 1:, 1:])
    m["sina"][:, 1:, 1:] = 0.5 * (m["sin_sg8"][:, :-1, :-1] + m["sin_sg6"][:, 1:, 1:])
    m["sina"][:, 0, :] = 1.0; m["sina"][:, :, 0] = 1.0
    m["cosa_s"] = m["cos_sg5"].copy()
    tiny = 1e-100
    m["rsin_u"] = 1.0 / np.maximum(tiny, m["sina_u"] ** 2)
    m["rsin_v"] = 1.0 / np.maximum(tiny, m["sina_v"] ** 2)
    m["rsin2"] = 1.0 / np.maximum(tiny, m["sin_sg5"] ** 2)
    m["rsina"] = 1.0 / np.maximum(tiny, m["sina"] ** 2)
    e1i, eni = NG, NG + n                                        # plane indices of i=1 and i=npx
    big = 1.0e30
    m["rsina"][:, [e1i, eni], :] = big; m["rsina"][:, :, [e1i, eni]] = big            # :527-538 (face edges)
    m["rsin_u"][:, :, [e1i, eni]] = 1.0 / m["sina_u"][:, :, [e1i, eni]]                 # :540-547
    m["rsin_v"][:, [e1i, eni], :] = 1.0 / m["sina_v"][:, [e1i, eni], :]                 # :549-556
    # divergence / del-n damping metrics (:700-725)
    m["divg_u"] = m["sina_v"] * m["dyc"] / m["dx"]; m["del6_u"] = m["sina_v"] * m["dx"] / m["dyc"]
    m["divg_v"] = m["sina_u"] * m["dxc"] / m["dy"]; m["del6_v"] = m["sina_u"] * m["dy"] / m["dxc"]
    for row in (e1i, eni):                                       # j = 1, npy: 0.5*(sin_sg(i,j,2)+sin_sg(i,j-1,4))
        f = 0.5 * (m["sin_sg2"][:, row, :] + m["sin_sg4"][:, row - 1, :])
        m["divg_u"][:, row, :] = f * m["dyc"][:, row, :] / m["dx"][:, row, :]
        m["del6_u"][:, row, :] = f * m["dx"][:, row, :] / m["dyc"][:, row, :]
    for col in (e1i, eni):
        f = 0.5 * (m["sin_sg1"][:, :, col] + m["sin_sg3"][:, :, col - 1])
        m["divg_v"][:, :, col] = f * m["dxc"][:, :, col] / m["dy"][:, :, col]
        m["del6_v"][:, :, col] = f * m["dy"][:, :, col] / m["dxc"][:, :, col]
    # Coriolis
    latc = np.arcsin(np.clip(ctr[..., 2], -1, 1)); latk = np.arcsin(np.clip(P[..., 2], -1, 1))
    m["f0"][sl] = 2.0 * omega * np.sin(latc); m["fC"] = 2.0 * omega * np.sin(latk)
    inner = (slice(None), slice(NG, NG + n), slice(NG, NG + n))
    da_min = float(m["area"][inner].min())
    da_min_c = float(area_c[:, NG:NG + n + 1, NG:NG + n + 1].min())
    # a2b_ord4 edge weights (fv_grid_utils_nlm.F90 edge_w/e/s/n: linear interpolation weights of the edge
    # mid-points to the corner) and extrap_corner coefficients x1/(x2-x1)
    edge = np.zeros((6, 4, pj))                                  # order w, e, s, n; index by plane position
    def ew(pa, pb, pc):                                          # corner pb between mid-points pa, pc: weight of pa
        d1, d2 = _dist(pa, pb), _dist(pb, pc)
        return d2 / (d1 + d2)
    for t in range(6):
        for j in range(2, n + 1):                                # qout(1,j) = edge_w(j)*q2(j-1) + (1-edge_w(j))*q2(j)
            jj = j + NG - 1
            edge[t, 0, jj] = ew(mw[t, jj - 1, e1i], P[t, jj, e1i], mw[t, jj, e1i])
            edge[t, 1, jj] = ew(mw[t, jj - 1, eni], P[t, jj, eni], mw[t, jj, eni])
            edge[t, 2, jj] = ew(ms[t, e1i, jj - 1], P[t, e1i, jj], ms[t, e1i, jj])
            edge[t, 3, jj] = ew(ms[t, eni, jj - 1], P[t, eni, jj], ms[t, eni, jj])
    ecorner = np.zeros((6, 4, 3))
    def ec(p0, p1, p2):
        x1, x2 = _dist(p1, p0), _dist(p2, p0)
        return x1 / (x2 - x1)
    C = lambda t, i, j: ctr[t, j + NG - 1, i + NG - 1]
    G = lambda t, i, j: P[t, j + NG - 1, i + NG - 1]
    npx = n + 1
    for t in range(6):
        ecorner[t, 0] = [ec(G(t, 1, 1), C(t, 1, 1), C(t, 2, 2)), ec(G(t, 1, 1), C(t, 0, 1), C(t, -1, 2)), ec(G(t, 1, 1), C(t, 1, 0), C(t, 2, -1))]
        ecorner[t, 1] = [ec(G(t, npx, 1), C(t, npx - 1, 1), C(t, npx - 2, 2)), ec(G(t, npx, 1), C(t, npx - 1, 0), C(t, npx - 2, -1)),
                         ec(G(t, npx, 1), C(t, npx, 1), C(t, npx + 1, 2))]
        ecorner[t, 2] = [ec(G(t, npx, npx), C(t, npx - 1, npx - 1), C(t, npx - 2, npx - 2)), ec(G(t, npx, npx), C(t, npx, npx - 1), C(t, npx + 1, npx - 2)),
                         ec(G(t, npx, npx), C(t, npx - 1, npx), C(t, npx - 2, npx + 1))]
        ecorner[t, 3] = [ec(G(t, 1, npx), C(t, 1, npx - 1), C(t, 2, npx - 2)), ec(G(t, 1, npx), C(t, 0, npx - 1), C(t, -1, npx - 2)),
                         ec(G(t, 1, npx), C(t, 1, npx), C(t, 2, npx + 1))]
    out = {k: np.ascontiguousarray(v) for k, v in m.items()}
    return out, da_min, da_min_c, edge, ecorner, dict(corners=P, centers=ctr)


def cube_fields(n, npz, geo, seed, kind="traj", opt=None):
    """Smooth global fields sampled on all six faces (halo included, so they are exchange-consistent):
    u, v (D-grid covariant winds), pt, delp as [6, npz, pj, pi]; kind="pert" gives zero-mean perturbations."""
    from .grid import hybrid_levels
    rng = np.random.default_rng(seed)
    P, ctr = geo["corners"], geo["centers"]
    pj = n + 2 * NG + 1
    nm = 4
    def scalar(pts, amp):
        f = np.zeros(pts.shape[:-1])
        for _ in range(nm):
            k = rng.standard_normal(3) * 2.0; ph = rng.uniform(0, 2 * np.pi)
            f += amp / nm * np.sin(pts @ k + ph)
        return f
    def vector(amp):
        cs = [(rng.standard_normal(3), rng.standard_normal(3) * 1.5, rng.uniform(0, 2 * np.pi)) for _ in range(nm)]
        def V(pts):
            out = np.zeros(pts.shape)
            for c, k, ph in cs:
                out += amp / nm * np.cross(c, pts) * np.cos(pts @ k + ph)[..., None]
            return out
        return V
    ctrp = np.zeros((6, pj, pj, 3)); ctrp[:, :-1, :-1] = ctr; ctrp[:, -1, :] = ctrp[:, -2, :]; ctrp[:, :, -1] = ctrp[:, :, -2]
    mid_y = np.zeros((6, pj, pj, 3)); tan_y = np.zeros((6, pj, pj, 3))     # south edge of cell (i,j): u point
    mid_y[:, :, :-1] = _norm(P[:, :, :-1] + P[:, :, 1:]); tan_y[:, :, :-1] = _tangent(mid_y[:, :, :-1], P[:, :, :-1], P[:, :, 1:])
    mid_x = np.zeros((6, pj, pj, 3)); tan_x = np.zeros((6, pj, pj, 3))     # west edge: v point
    mid_x[:, :-1, :] = _norm(P[:, :-1, :] + P[:, 1:, :]); tan_x[:, :-1, :] = _tangent(mid_x[:, :-1, :], P[:, :-1, :], P[:, 1:, :])
    mid_y[:, :, -1] = mid_y[:, :, -2]; mid_x[:, -1, :] = mid_x[:, -2, :]
    out = {n_: np.zeros((6, npz, pj, pj)) for n_ in ("u", "v", "pt", "delp")}
    if kind == "traj":
        ak, bk = hybrid_levels(npz, opt.ptop)
        ps = 1.0e5 + scalar(ctrp, 800.0)
        V0 = vector(25.0)
        for k in range(npz):
            out["delp"][:, k] = (ak[k + 1] - ak[k]) + (bk[k + 1] - bk[k]) * ps
        pe = np.concatenate([np.full((6, 1, pj, pj), opt.ptop), opt.ptop + np.cumsum(out["delp"], axis=1)], axis=1)
        pmid = 0.5 * (pe[:, 1:] + pe[:, :-1])
        for k in range(npz):
            T = 288.0 * (np.maximum(pmid[:, k], 2.0e4) / 1.0e5) ** 0.19 + scalar(ctrp, 2.0)
            out["pt"][:, k] = T / pmid[:, k] ** opt.akap
            Vk = vector(6.0)
            fac = 0.6 + 0.8 * k / max(1, npz - 1)
            out["u"][:, k] = np.einsum("...i,...i", fac * V0(mid_y) + Vk(mid_y), tan_y)
            out["v"][:, k] = np.einsum("...i,...i", fac * V0(mid_x) + Vk(mid_x), tan_x)
        phis = 9.80 * 400.0 * (1.0 + scalar(ctrp, 1.0))
        return out, phis, ak, bk
    for k in range(npz):
        out["delp"][:, k] = scalar(ctrp, 10.0); out["pt"][:, k] = scalar(ctrp, 0.02)
        Vk = vector(1.0)
        out["u"][:, k] = np.einsum("...i,...i", Vk(mid_y), tan_y); out["v"][:, k] = np.einsum("...i,...i", Vk(mid_x), tan_x)
    return out


# --------------------------------------------------------------------------------- faces over ranks
def faces_of(rank, world, ntiles=6):
    """Tiles (0-based; the six faces when ntiles = 6) owned by `rank` when they are dealt over `world` ranks in contiguous blocks
    (6/3/2+1 faces per GPU at 1/2/4 GPUs, one each at 6, ranks >= 6 idle; 24 tiles of a 2 x 2 layout: 3 per GPU at 8: SURVEY.md 8e)."""
    if world >= ntiles:
        return [rank] if rank < ntiles else []
    base, extra = divmod(ntiles, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def face_owner(world, ntiles=6):
    own = {}
    for r in range(world):
        for f in faces_of(r, world, ntiles):
            own[f] = r
    return own


def split_table(tab, rank, world, ntiles=6, loopback=False):
    """One exchange table of the whole cube -> what `rank` needs: (local rows [n,7] with local tile numbers,
    peers, {peer: send rows [m,3] = field, local tile, index}, {peer: recv rows [m,4] = field, local tile, index, sign}).
    Both ends walk the global table in the same order, so message layouts agree without negotiation.
    loopback: the rows between two tiles of this rank travel as messages too, with the rank as its own peer (pack -> send to
    self / receive from self -> unpack) — the message path exercised end to end where only one GPU is at hand."""
    own = face_owner(world, ntiles)
    mine = faces_of(rank, world, ntiles)
    loc = {f: n for n, f in enumerate(mine)}
    local, send, recv = [], {}, {}
    for r in tab:
        do, so = own[int(r[1])], own[int(r[4])]
        if do == rank and so == rank and loopback:
            recv.setdefault(rank, []).append((r[0], loc[int(r[1])], r[2], r[6]))
            send.setdefault(rank, []).append((r[3], loc[int(r[4])], r[5]))
        elif do == rank and so == rank:
            local.append((r[0], loc[int(r[1])], r[2], r[3], loc[int(r[4])], r[5], r[6]))
        elif do == rank:
            recv.setdefault(so, []).append((r[0], loc[int(r[1])], r[2], r[6]))
        elif so == rank:
            send.setdefault(do, []).append((r[3], loc[int(r[4])], r[5]))
    peers = sorted(set(send) | set(recv))
    A = lambda l, w: np.array(l, dtype=np.int32).reshape(-1, w)
    return A(local, 7), peers, {p: A(send.get(p, []), 3) for p in peers}, {p: A(recv.get(p, []), 4) for p in peers}

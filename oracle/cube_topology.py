"""TEST INFRASTRUCTURE -- an independent derivation of the five halo-exchange tables of include/fv3lm.h, used only by
tests/test_cube.py to break the common mode between oracle and product (both apply the tables of
fv3_jedi_linearmodel_amd/cube.py, which derives them geometrically from six face frames on the unit cube).

This module knows nothing about frames or geometry and imports nothing from the package.  Its only input is the contact
list the reference hands to mpp_define_mosaic (TOOLS/fv_mp_nlm_mod.F90:524-572, transcribed below as data: tile pair and the
start/end indices of the two cell lines in contact).  Everything else is index arithmetic in the spirit of
mpp_update_domains: each contact defines an affine map of index space from "beyond edge E1 of tile 1" to "inside edge E2 of
tile 2" (along-edge coordinate start->end on both sides, depth beyond the edge -> depth inside the neighbour); halo points are
pushed through the map of the edge they lie beyond, vector components through its linear part (a signed permutation: that is
the D-/C-grid component swap and sign flip of rotated contacts).  Row format and field conventions as cube.exchange_table."""
import numpy as np

NG = 3
# (tile1, istart1, iend1, jstart1, jend1, tile2, istart2, iend2, jstart2, jend2) with "n" = nx = ny  (fv_mp_nlm_mod.F90:524-572)
CONTACT_LINES = [
    (1, "n", "n", 1, "n", 2, 1, 1, 1, "n"), (1, 1, "n", "n", "n", 3, 1, 1, "n", 1), (1, 1, 1, 1, "n", 5, "n", 1, "n", "n"),
    (1, 1, "n", 1, 1, 6, 1, "n", "n", "n"), (2, 1, "n", "n", "n", 3, 1, "n", 1, 1), (2, "n", "n", 1, "n", 4, "n", 1, 1, 1),
    (2, 1, "n", 1, 1, 6, "n", "n", "n", 1), (3, "n", "n", 1, "n", 4, 1, 1, 1, "n"), (3, 1, "n", "n", "n", 5, 1, 1, "n", 1),
    (4, 1, "n", "n", "n", 5, 1, "n", 1, 1), (4, "n", "n", 1, "n", 6, "n", 1, 1, 1), (5, "n", "n", 1, "n", 6, 1, 1, 1, "n")]
LOC = {"center": (0.5, 0.5), "corner": (0.0, 0.0), "yedge": (0.5, 0.0), "xedge": (0.0, 0.5)}
FIELDS = {"cell": [("center", "s")], "dvec": [("yedge", "1"), ("xedge", "2")], "cvec": [("xedge", "1"), ("yedge", "2")],
          "corner": [("corner", "s")]}


def _line(n, i0, i1, j0, j1):
    """a line of n cells hugging one edge -> (edge name, start corner, unit along-direction, outward normal) in index space,
    where cell (i,j) covers [i-1,i] x [j-1,j] and the tile [0,n]^2"""
    v = lambda a: n if a == "n" else a
    i0, i1, j0, j1 = v(i0), v(i1), v(j0), v(j1)
    if i0 == i1:                        # west or east edge, traversed in j
        x = 0.0 if i0 == 1 else float(n); out = np.array([-1.0, 0.0]) if i0 == 1 else np.array([1.0, 0.0])
        up = j1 > j0
        return ("W" if i0 == 1 else "E"), np.array([x, 0.0 if up else float(n)]), np.array([0.0, 1.0 if up else -1.0]), out
    y = 0.0 if j0 == 1 else float(n); out = np.array([0.0, -1.0]) if j0 == 1 else np.array([0.0, 1.0])
    up = i1 > i0
    return ("S" if j0 == 1 else "N"), np.array([0.0 if up else float(n), y]), np.array([1.0 if up else -1.0, 0.0]), out


def edge_maps(n):
    """{(tile, edge): (neighbour tile, J, b)}: a point p beyond that edge is p' = J p + b on the neighbour"""
    maps = {}
    for (t1, a, b, c, d, t2, e, f, g, h) in CONTACT_LINES:
        E1, s1, u1, o1 = _line(n, a, b, c, d)
        E2, s2, u2, o2 = _line(n, e, f, g, h)
        for (ta, Ea, sa, ua, oa, tb, sb, ub, ob) in ((t1, E1, s1, u1, o1, t2, s2, u2, o2), (t2, E2, s2, u2, o2, t1, s1, u1, o1)):
            # along ua -> along ub, outward oa -> inward -ob
            A = np.stack([ua, oa], axis=1); B = np.stack([ub, -ob], axis=1)
            J = B @ np.linalg.inv(A)
            maps[(ta, Ea)] = (tb, np.rint(J), sb - J @ sa)
    assert len(maps) == 24
    return maps


def _beyond(x, y, n):
    ox, oy = (x < 0) or (x > n), (y < 0) or (y > n)
    if ox and oy:
        return "corner"
    if ox:
        return "W" if x < 0 else "E"
    if oy:
        return "S" if y < 0 else "N"
    return None


def plane_index(i, j, n):
    return (j + NG - 1) * (n + 2 * NG + 1) + (i + NG - 1)


def _row(n, maps, tile, fields, fi, i, j, x, y, edge):
    kind, dirn = fields[fi]
    tb, J, b = maps[(tile, edge)]
    xb, yb = J @ np.array([x, y]) + b
    rotated = J[0, 0] == 0
    kb = {"yedge": "xedge", "xedge": "yedge"}.get(kind, kind) if rotated else kind
    if dirn == "s":
        sf, sign = fi, 1
    else:
        col = J[:, 0 if dirn == "1" else 1]
        cb = int(np.argmax(np.abs(col))); sign = int(col[cb])
        sf = fields.index((kb, "1" if cb == 0 else "2"))
    oxb, oyb = LOC[kb]
    ib, jb = int(round(xb - oxb)) + 1, int(round(yb - oyb)) + 1
    assert abs(ib - 1 + oxb - xb) < 1e-4 and abs(jb - 1 + oyb - yb) < 1e-4, (tile, i, j, xb, yb)
    return (fi, tile - 1, plane_index(i, j, n), sf, tb - 1, plane_index(ib, jb, n), sign)


def exchange_table(n, kind):
    maps = edge_maps(n)
    fields = FIELDS[kind]
    rows = []
    lo, hi = 1 - NG, n + NG
    for tile in range(1, 7):
        for fi, (k, _) in enumerate(fields):
            ox, oy = LOC[k]
            for j in range(lo, hi + (1 if oy == 0.0 else 0) + 1):
                for i in range(lo, hi + (1 if ox == 0.0 else 0) + 1):
                    x, y = i - 1 + ox, j - 1 + oy
                    e = _beyond(x, y, n)
                    if e is None or e == "corner":
                        continue
                    rows.append(_row(n, maps, tile, fields, fi, i, j, x, y, e))
    return np.array(rows, dtype=np.int32)


def boundary_table(n):
    """mpp_get_boundary of (u, v): u(1:n, n+1) from across the north edge, v(n+1, 1:n) from across the east edge"""
    maps = edge_maps(n)
    fields = FIELDS["dvec"]
    rows, eps = [], 1.0e-6
    for tile in range(1, 7):
        for i in range(1, n + 1):
            rows.append(_row(n, maps, tile, fields, 0, i, n + 1, i - 0.5, n + eps, "N"))
        for j in range(1, n + 1):
            rows.append(_row(n, maps, tile, fields, 1, n + 1, j, n + eps, j - 0.5, "E"))
    return np.array(rows, dtype=np.int32)


def all_tables(n):
    t = {k: exchange_table(n, k) for k in FIELDS}
    t["dedge"] = boundary_table(n)
    return t


# ---------------------------------------------------------------------------------------------------------------------------
# Metric terms by other formulas.  Input: the cell-corner unit vectors of a face (what a host hands over as grid(:,:,1:2));
# output: dx, dy, dxa, dya, area and the centre angle (sin_sg5, cos_sg5) computed in longitude / latitude with textbook
# spherical trigonometry (haversine distance, l'Huilier's spherical excess, initial bearings) -- none of the chord / triple-product /
# tangent-projection expressions of cube.cubed_sphere_metrics.  Definitions: NLM/fv_grid_utils_nlm.F90:455-745.
def _lonlat(p):
    return np.arctan2(p[..., 1], p[..., 0]), np.arcsin(np.clip(p[..., 2], -1.0, 1.0))


def _haversine(a, b):
    (l1, p1), (l2, p2) = _lonlat(a), _lonlat(b)
    h = np.sin(0.5 * (p2 - p1)) ** 2 + np.cos(p1) * np.cos(p2) * np.sin(0.5 * (l2 - l1)) ** 2
    return 2.0 * np.arcsin(np.sqrt(np.clip(h, 0.0, 1.0)))


def _mid(a, b):
    m = a + b
    return m / np.sqrt(np.sum(m * m, axis=-1, keepdims=True))


def _excess(a, b, c):
    """spherical excess of a triangle from its side lengths (l'Huilier)"""
    A, B, C = _haversine(b, c), _haversine(a, c), _haversine(a, b)
    s = 0.5 * (A + B + C)
    t = np.tan(0.5 * s) * np.tan(0.5 * (s - A)) * np.tan(0.5 * (s - B)) * np.tan(0.5 * (s - C))
    return 4.0 * np.arctan(np.sqrt(np.maximum(t, 0.0)))


def _bearing(a, b):
    """initial bearing (clockwise from north) of the great circle from a to b"""
    (l1, p1), (l2, p2) = _lonlat(a), _lonlat(b)
    return np.arctan2(np.sin(l2 - l1) * np.cos(p2), np.cos(p1) * np.sin(p2) - np.sin(p1) * np.cos(p2) * np.cos(l2 - l1))


def face_metrics_lonlat(P, radius):
    """P: [nj+1, ni+1, 3] corner unit vectors of cells (j, i).  Returns dict of [nj, ni] (dx: [nj+1, ni], dy: [nj, ni+1])."""
    c00, c10, c01, c11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    mw, me, ms, mn = _mid(c00, c01), _mid(c10, c11), _mid(c00, c10), _mid(c01, c11)
    ctr = c00 + c10 + c01 + c11; ctr = ctr / np.sqrt(np.sum(ctr * ctr, axis=-1, keepdims=True))
    out = {"dx": radius * _haversine(P[:, :-1], P[:, 1:]), "dy": radius * _haversine(P[:-1, :], P[1:, :]),
           "dxa": radius * _haversine(mw, me), "dya": radius * _haversine(ms, mn),
           "area": radius ** 2 * (_excess(c00, c10, c11) + _excess(c00, c11, c01))}
    # angle at the centre between the direction west-mid -> east-mid and south-mid -> north-mid
    ang = _bearing(ctr, me) - _bearing(ctr, mn)
    out["cos_sg5"], out["sin_sg5"] = np.cos(ang), np.abs(np.sin(ang))
    return out

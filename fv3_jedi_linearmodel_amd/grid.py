"""Host-side metric terms and synthetic states for stand-alone runs (tests, bench).

In production the metric arrays come from the Fortran host (fv_grid_type, filled by
grid_utils_init, NLM/fv_grid_utils_nlm.F90:78-840) and are only uploaded.  For stand-alone runs this
module builds a smooth, doubly-periodic, non-orthogonal tile whose derived terms obey the same
algebraic relations as the reference's (rsin_u = 1/sina_u**2 :503, rsin2 = 1/sin_sg5**2 :519,
rsina = 1/sina**2 :536, divg_u = sina_v*dyc/dx :708, del6_v = sina_u*dy/dxc :716, ...).
"""
import numpy as np

METRIC_NAMES = ("area,rarea,rarea_c,dx,dy,dxa,dya,dxc,dyc,rdx,rdy,rdxa,rdya,rdxc,rdyc,cosa,sina,rsina,cosa_u,cosa_v,"
                "cosa_s,sina_u,sina_v,rsin_u,rsin_v,rsin2,f0,fC,del6_u,del6_v,divg_u,divg_v,"
                "sin_sg1,sin_sg2,sin_sg3,sin_sg4,sin_sg5,sin_sg6,sin_sg7,sin_sg8,sin_sg9,"
                "cos_sg1,cos_sg2,cos_sg3,cos_sg4,cos_sg5,cos_sg6,cos_sg7,cos_sg8,cos_sg9").split(",")
NG = 3


def _coords(nx, ny):
    """index coordinates of the padded plane: x[i] for i = isd..ied+1 (corner positions)."""
    i = np.arange(1 - NG, nx + NG + 2, dtype=np.float64)
    j = np.arange(1 - NG, ny + NG + 2, dtype=np.float64)
    return np.meshgrid(i, j, indexing="xy")  # shape (pj, pi)


def synthetic_tile_metrics(nx, ny, dx0=2.0e5, skew=0.25, stretch=0.08, lat0=35.0, ntile=1):
    """Smooth doubly-periodic tile.  Returns (dict name -> (ntile,pj,pi) array, da_min, da_min_c)."""
    X, Y = _coords(nx, ny)          # corner (i, j)
    tx, ty = 2 * np.pi / nx, 2 * np.pi / ny

    def DX(x, y):
        return dx0 * (1.0 + stretch * np.sin(tx * (x - 1)) * np.cos(ty * (y - 1)))

    def DY(x, y):
        return dx0 * (1.0 - stretch * np.cos(tx * (x - 1)) * np.sin(ty * (y - 1)))

    def ALPHA(x, y):               # angle between the coordinate lines
        return 0.5 * np.pi - skew * np.sin(tx * (x - 1) + 0.3) * np.sin(ty * (y - 1) - 0.2)

    m = {}
    m["dx"] = DX(X + 0.5, Y); m["dy"] = DY(X, Y + 0.5)
    m["dxa"] = DX(X + 0.5, Y + 0.5); m["dya"] = DY(X + 0.5, Y + 0.5)
    m["dxc"] = DX(X, Y + 0.5); m["dyc"] = DY(X + 0.5, Y)
    for n in ("dx", "dy", "dxa", "dya", "dxc", "dyc"):
        m["r" + n] = 1.0 / m[n]
    ac, au, av, as_ = ALPHA(X, Y), ALPHA(X, Y + 0.5), ALPHA(X + 0.5, Y), ALPHA(X + 0.5, Y + 0.5)
    m["cosa"], m["sina"] = np.cos(ac), np.sin(ac)
    m["cosa_u"], m["sina_u"] = np.cos(au), np.sin(au)
    m["cosa_v"], m["sina_v"] = np.cos(av), np.sin(av)
    m["cosa_s"] = np.cos(as_)
    m["rsina"] = 1.0 / m["sina"] ** 2
    m["rsin_u"] = 1.0 / m["sina_u"] ** 2
    m["rsin_v"] = 1.0 / m["sina_v"] ** 2
    # sin_sg/cos_sg(i,j,n): n=1 west edge mid-point, 2 south, 3 east, 4 north, 5 centre, 6..9 corners
    pos = {1: (X, Y + 0.5), 2: (X + 0.5, Y), 3: (X + 1.0, Y + 0.5), 4: (X + 0.5, Y + 1.0), 5: (X + 0.5, Y + 0.5),
           6: (X, Y), 7: (X + 1.0, Y), 8: (X + 1.0, Y + 1.0), 9: (X, Y + 1.0)}
    for n, (x, y) in pos.items():
        a = ALPHA(x, y)
        m["sin_sg%d" % n], m["cos_sg%d" % n] = np.sin(a), np.cos(a)
    m["rsin2"] = 1.0 / m["sin_sg5"] ** 2
    m["area"] = m["dxa"] * m["dya"] * m["sin_sg5"]
    m["rarea"] = 1.0 / m["area"]
    area_c = DX(X, Y) * DY(X, Y) * m["sina"]
    m["rarea_c"] = 1.0 / area_c
    omega = 7.292e-5
    lat_c = np.deg2rad(lat0 + 10.0 * np.sin(ty * (Y + 0.5 - 1)))
    lat_k = np.deg2rad(lat0 + 10.0 * np.sin(ty * (Y - 1)))
    m["f0"] = 2.0 * omega * np.sin(lat_c) + 0.0 * X
    m["fC"] = 2.0 * omega * np.sin(lat_k) + 0.0 * X
    m["divg_u"] = m["sina_v"] * m["dyc"] / m["dx"]
    m["del6_u"] = m["sina_v"] * m["dx"] / m["dyc"]
    m["divg_v"] = m["sina_u"] * m["dxc"] / m["dy"]
    m["del6_v"] = m["sina_u"] * m["dy"] / m["dxc"]
    sl = (slice(NG, NG + ny), slice(NG, NG + nx))
    da_min = float(m["area"][sl].min())
    da_min_c = float(area_c[sl].min())
    out = {n: np.ascontiguousarray(np.broadcast_to(m[n], (ntile,) + m[n].shape)) for n in METRIC_NAMES}
    return out, da_min, da_min_c


def hybrid_levels(npz, ptop=1.0, p0=1.0e5):
    """Smooth synthetic hybrid coefficients: pe(k) = ak(k) + bk(k)*ps, pure pressure aloft."""
    s = np.linspace(0.0, 1.0, npz + 1) ** 1.6
    bk = (np.maximum(s - 0.15, 0.0) / 0.85) ** 1.3
    ak = ptop + s * (p0 - ptop) - bk * p0
    return ak, bk


def _smooth_field(rng, shape, nx, ny, amp, nmodes=3):
    """Doubly-periodic smooth random field on the padded plane, level-dependent."""
    ntile, nk, pj, pi = shape
    X, Y = _coords(nx, ny)
    f = np.zeros(shape)
    for t in range(ntile):
        for k in range(nk):
            a = np.zeros((pj, pi))
            for _ in range(nmodes):
                kx, ky = rng.integers(1, 3), rng.integers(1, 3)
                ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
                a += rng.normal() * np.sin(2 * np.pi * kx * (X - 1) / nx + ph1) * np.cos(2 * np.pi * ky * (Y - 1) / ny + ph2)
            f[t, k] = amp * a / np.sqrt(nmodes)
    return f


def synthetic_state(nx, ny, npz, opt, seed=20250114, ntile=1):
    """Trajectory (u, v, pt, delp, phis, ak, bk) of SURVEY.md §8d flavour on the padded plane:
    20 m/s flow + waves, isothermal-ish theta_v profile, delp from ak/bk with ps = 1000 hPa +- 5 hPa.
    pt is the dycore's internal variable (virtual potential temperature / pkz scaling, cp*... no:
    pt = T_v / pkz as in fv_dynamics_tlm.F90:583-595)."""
    rng = np.random.default_rng(seed)
    ak, bk = hybrid_levels(npz, opt.ptop)
    X, Y = _coords(nx, ny)
    shp = (ntile, npz, ny + 7, nx + 7)
    ps = 1.0e5 + 500.0 * np.sin(2 * np.pi * (X - 1) / nx) * np.cos(2 * np.pi * (Y - 1) / ny)
    delp = np.empty(shp)
    for k in range(npz):
        delp[:, k] = (ak[k + 1] - ak[k]) + (bk[k + 1] - bk[k]) * ps
    pe = np.concatenate([np.full((ntile, 1) + shp[2:], opt.ptop), opt.ptop + np.cumsum(delp, axis=1)], axis=1)
    pmid = 0.5 * (pe[:, 1:] + pe[:, :-1])
    T = 288.0 * (np.maximum(pmid, 2.0e4) / 1.0e5) ** 0.19 + _smooth_field(rng, shp, nx, ny, 1.5)
    pt = T / pmid ** opt.akap
    u = 20.0 + 5.0 * np.cos(2 * np.pi * (Y - 1) / ny)[None, None] + _smooth_field(rng, shp, nx, ny, 3.0)
    v = 3.0 + 5.0 * np.sin(2 * np.pi * (X - 1) / nx)[None, None] + _smooth_field(rng, shp, nx, ny, 3.0)
    phis = 9.80 * 500.0 * np.exp(-(((X - nx / 2) / (0.2 * nx)) ** 2 + ((Y - ny / 2) / (0.2 * ny)) ** 2))
    phis = np.broadcast_to(phis, (ntile,) + phis.shape).copy()
    # make the hill periodic-consistent in the halo
    return dict(u=u, v=v, pt=pt, delp=delp), phis, ak, bk


def synthetic_pert(nx, ny, npz, seed=20250115, ntile=1, scale=None):
    """Smooth perturbation (1 m/s, 1 m/s, theta ~ 1 K equivalent, 10 Pa)."""
    rng = np.random.default_rng(seed)
    shp = (ntile, npz, ny + 7, nx + 7)
    sc = scale or dict(u=1.0, v=1.0, pt=0.02, delp=10.0)
    return {n: _smooth_field(rng, shp, nx, ny, sc[n]) for n in ("u", "v", "pt", "delp")}


def halo_fill_periodic(a, nx, ny):
    """Wrap every point of the padded plane outside 1..nx x 1..ny (same rule as the device halo)."""
    pj, pi = a.shape[-2:]
    ii = (np.arange(pi) - NG) % nx + NG
    jj = (np.arange(pj) - NG) % ny + NG
    return np.ascontiguousarray(a[..., jj[:, None], ii[None, :]])

"""Seeded synthetic Waymo-shaped scenes (no dataset, no network).

Stands in for ``WaymoDataset.__getitem__`` (seg3d/datasets/waymo_dataset.py:281-336)
on the bench / test path: produces the raw per-sample point rows the reference
feeds to its voxelizer -- ``[x, y, z, time_lag(=0), tanh(intensity), elongation]``
float32 (waymo_dataset.py:150-153) -- with the TOP-lidar geometry of
tools/waymo_parser.py:14-15 (64 beams x 2650 azimuth columns).

Geometry: sensor 2 m above a ground plane at z=0, beam inclinations uniformly
spaced in [-17.6 deg, +2.4 deg]; each of 72 azimuth sectors has a vertical wall
at distance U(8, 70) m; returns are the nearer of ground and wall, clipped at
75 m, with N(0, 2 cm) range noise; ~10 k short-range "side lidar" points are
added inside 20 m.  About 180 k points per scene.
"""
import numpy as np

N_BEAMS = 64
N_AZIMUTH = 2650
SENSOR_HEIGHT = 2.0
MAX_RANGE = 75.0


def make_scene(seed, n_side=10000, dtype=np.float32):
    """One sweep, rows [x, y, z, 0, tanh(intensity), elongation] (N, 6)."""
    rs = np.random.RandomState(1000 + int(seed))
    inc = np.deg2rad(np.linspace(-17.6, 2.4, N_BEAMS))
    az = np.linspace(-np.pi, np.pi, N_AZIMUTH, endpoint=False)
    az = az + rs.uniform(0, 2 * np.pi / N_AZIMUTH)
    inc_g, az_g = np.meshgrid(inc, az, indexing="ij")

    n_sector = 72
    wall = rs.uniform(8.0, 70.0, size=n_sector)
    sector = ((az_g + np.pi) / (2 * np.pi) * n_sector).astype(np.int64) % n_sector
    d_wall = wall[sector] / np.cos(inc_g)  # slant range to the wall
    with np.errstate(divide="ignore"):
        d_ground = np.where(inc_g < 0, SENSOR_HEIGHT / np.sin(-inc_g), np.inf)
    rng = np.minimum(d_wall, d_ground)
    rng = rng + rs.normal(0.0, 0.02, size=rng.shape)
    keep = (rng > 1.0) & (rng < MAX_RANGE)
    # a few percent of beams give no return
    keep &= rs.uniform(size=rng.shape) > 0.03
    r, i_, a_ = rng[keep], inc_g[keep], az_g[keep]
    xyz = np.stack([r * np.cos(i_) * np.cos(a_), r * np.cos(i_) * np.sin(a_),
                    SENSOR_HEIGHT + r * np.sin(i_)], axis=1)

    # side lidars / second returns: short range, wide vertical field of view
    rs_r = rs.uniform(1.0, 20.0, size=n_side)
    rs_a = rs.uniform(-np.pi, np.pi, size=n_side)
    rs_z = np.abs(rs.normal(0.0, 0.6, size=n_side))
    side = np.stack([rs_r * np.cos(rs_a), rs_r * np.sin(rs_a), rs_z], axis=1)
    xyz = np.concatenate([xyz, side], axis=0)

    n = xyz.shape[0]
    feat = np.stack([np.zeros(n), np.tanh(rs.uniform(0, 1, size=n)), rs.uniform(0, 1, size=n)], axis=1)
    return np.concatenate([xyz, feat], axis=1).astype(dtype)


def cart2polar_rows(points):
    """Cylinder-config input rows (waymo_dataset.py:270-273 with pointops_utils.py:8-11):
    [rho, phi, z, x, y, feat3..5] -> (N, 8)."""
    rho = np.sqrt(points[:, 0] ** 2 + points[:, 1] ** 2)
    phi = np.arctan2(points[:, 1], points[:, 0])
    polar = np.stack((rho, phi, points[:, 2]), axis=1)
    return np.concatenate((polar, points[:, :2], points[:, 3:]), axis=1).astype(points.dtype)


def make_small_scene(seed, n_points, extent=12.0, dtype=np.float32):
    """Small clustered scene for parity tests: points on a few planes/blobs inside
    +-extent metres so that windows at every stage hold several voxels."""
    rs = np.random.RandomState(5000 + int(seed))
    n_ground = n_points // 2
    g = np.stack([rs.uniform(-extent, extent, n_ground), rs.uniform(-extent, extent, n_ground),
                  rs.normal(0.0, 0.03, n_ground)], axis=1)
    n_wall = n_points // 4
    w = np.stack([rs.uniform(-extent, extent, n_wall), np.full(n_wall, 0.37 * extent) + rs.normal(0, 0.03, n_wall),
                  rs.uniform(0.0, 3.0, n_wall)], axis=1)
    n_blob = n_points - n_ground - n_wall
    c = rs.uniform(-extent * 0.8, extent * 0.8, size=(6, 2))
    which = rs.randint(0, 6, n_blob)
    b = np.concatenate([c[which] + rs.normal(0, 0.5, (n_blob, 2)), np.abs(rs.normal(0.8, 0.5, (n_blob, 1)))], axis=1)
    xyz = np.concatenate([g, w, b], axis=0)
    xyz = xyz[rs.permutation(xyz.shape[0])]
    n = xyz.shape[0]
    feat = np.stack([np.zeros(n), np.tanh(rs.uniform(0, 1, size=n)), rs.uniform(0, 1, size=n)], axis=1)
    return np.concatenate([xyz, feat], axis=1).astype(dtype)


def make_multi_sweep_scene(seed, n_sweeps=3, dtype=np.float32):
    """Current sweep + (n_sweeps - 1) history sweeps re-posed by a small ego motion, time lag in column 3
    (waymo_dataset.py:156-202 merges sweeps this way; configs/waymo_multi_sweeps.yaml NUM_SWEEPS = 3).
    Returns (rows [N, 6], n_current): current-sweep rows come first and have time lag exactly 0."""
    rs = np.random.RandomState(9000 + int(seed))
    cur = make_scene(seed, dtype=np.float64)
    sweeps = [cur]
    for i in range(1, n_sweeps):
        hist = make_scene(seed, n_side=6000, dtype=np.float64)
        yaw = rs.normal(0.0, 0.01) * i
        c, s_ = np.cos(yaw), np.sin(yaw)
        xy = hist[:, :2] @ np.array([[c, -s_], [s_, c]]) + np.array([0.6 * i, rs.normal(0, 0.05)])
        hist[:, :2] = xy
        hist[:, :3] += rs.normal(0.0, 0.01, size=(hist.shape[0], 3))
        hist[:, 3] = 0.1 * i
        sweeps.append(hist)
    return np.concatenate(sweeps, axis=0).astype(dtype), cur.shape[0]


def make_image_features(seed, n_points, dim=28, hit_ratio=0.5, dtype=np.float32):
    """Per-point image features of tools/extract_image_feature.py shape: zero rows where no camera sees the point."""
    rs = np.random.RandomState(12000 + int(seed))
    f = rs.randn(n_points, dim).astype(dtype)
    f[rs.rand(n_points) >= hit_ratio] = 0
    return f


# ---------------------------------------------------------------------------------------------- BASELINE configs[4]
# "synthetic dense scene 2M pts @0.02m voxel" (SURVEY 8d Config 5): a 28.8 m x 28.8 m x 1.28 m block at 0.02 m voxels is the
# SAME 1440 x 1440 x 64 grid as the Waymo configs, so the model, the window geometry and every index width are unchanged
# -- only the number of active sites grows by an order of magnitude (the HBM-bound stress the config asks for).
DENSE_VOXEL = (0.02, 0.02, 0.02)
DENSE_RANGE = (-14.4, -14.4, -0.28, 14.4, 14.4, 1.0)


def make_dense_scene(seed, n_points=2_000_000, dtype=np.float32):
    """Close-range dense scan, rows [x, y, z, 0, tanh(intensity), elongation] (n_points, 6): an undulating floor sampled on
    a jittered raster (consecutive rows are spatial neighbours, as in a scan), vertical panels, and box-shaped clutter.
    About two points per occupied 2 cm voxel."""
    rs = np.random.RandomState(20000 + int(seed))
    lo, hi = np.array(DENSE_RANGE[:3]), np.array(DENSE_RANGE[3:])
    n_floor, n_wall = int(0.55 * n_points), int(0.30 * n_points)
    n_box = n_points - n_floor - n_wall
    # floor: raster of side ~ sqrt(n) with jitter, height = two sinusoids (+-6 cm) + 3 mm noise
    side = int(np.ceil(np.sqrt(n_floor)))
    iy, ix = np.divmod(np.arange(n_floor), side)
    pitch = (hi[0] - lo[0]) / side
    fx = lo[0] + (ix + rs.uniform(0, 1, n_floor)) * pitch
    fy = lo[1] + (iy + rs.uniform(0, 1, n_floor)) * pitch
    ph = rs.uniform(0, 2 * np.pi, 2)
    fz = 0.06 * np.sin(fx * 0.9 + ph[0]) * np.cos(fy * 0.7 + ph[1]) + rs.normal(0, 0.003, n_floor)
    floor = np.stack([fx, fy, fz], axis=1)
    # panels: 40 vertical rectangles, 0 .. 0.95 m high, 3 mm range noise, scanned row by row
    n_panels = 40
    per = n_wall // n_panels
    walls = []
    for _ in range(n_panels):
        c = rs.uniform(lo[:2] + 2, hi[:2] - 2)
        ang, length = rs.uniform(0, np.pi), rs.uniform(1.5, 6.0)
        rows = max(int(np.sqrt(per * 0.95 / length)), 1)
        cols = per // rows
        t = (np.arange(rows * cols) % cols + rs.uniform(0, 1, rows * cols)) / cols * length - length / 2
        h = (np.arange(rows * cols) // cols + rs.uniform(0, 1, rows * cols)) / rows * 0.95
        off = rs.normal(0, 0.003, rows * cols)
        walls.append(np.stack([c[0] + t * np.cos(ang) - off * np.sin(ang), c[1] + t * np.sin(ang) + off * np.cos(ang), h], axis=1))
    wall = np.concatenate(walls, axis=0)
    # clutter: points on the surfaces of 300 small boxes standing on the floor
    n_box = n_points - n_floor - wall.shape[0]
    centers = rs.uniform(lo[:2] + 1, hi[:2] - 1, size=(300, 2))
    sizes = rs.uniform(0.1, 0.6, size=(300, 3))
    which = np.sort(rs.randint(0, 300, n_box))
    u = rs.uniform(-0.5, 0.5, size=(n_box, 3))
    face = rs.randint(0, 3, n_box)
    u[np.arange(n_box), face] = np.sign(u[np.arange(n_box), face]) * 0.5  # snap one coordinate to a face
    box = np.concatenate([centers[which] + u[:, :2] * sizes[which, :2], (u[:, 2:] + 0.5) * sizes[which, 2:]], axis=1)
    xyz = np.concatenate([floor, wall, box], axis=0)
    xyz = np.clip(xyz, lo + 1e-3, hi - 1e-3)
    n = xyz.shape[0]
    feat = np.stack([np.zeros(n), np.tanh(rs.uniform(0, 1, size=n)), rs.uniform(0, 1, size=n)], axis=1)
    return np.concatenate([xyz, feat], axis=1).astype(dtype)

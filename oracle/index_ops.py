"""ctypes front-end of oracle/seg3d_oracle.c (TEST INFRASTRUCTURE ONLY).

numpy in, numpy out.  The shared object is built by ``make -C oracle`` (also run
by ``__graft_entry__.build()``); it is built on demand here if missing and gcc
is available.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libseg3d_oracle.so")
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(_HERE, "seg3d_oracle.c")
    if not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(_SO)
    i64, p = ctypes.c_int64, ctypes.c_void_p
    lib.oracle_voxelize_f32.restype = i64
    lib.oracle_voxelize_f32.argtypes = [p, i64, i64, p, p, p, p]
    lib.oracle_voxelize_f64.restype = i64
    lib.oracle_voxelize_f64.argtypes = [p, i64, i64, p, p, p, p]
    lib.oracle_ingroup_rank.restype = ctypes.c_int
    lib.oracle_ingroup_rank.argtypes = [p, i64, p]
    lib.oracle_rulebook_subm.restype = ctypes.c_int
    lib.oracle_rulebook_subm.argtypes = [p, i64, p, p]
    lib.oracle_downsample_coords.restype = i64
    lib.oracle_downsample_coords.argtypes = [p, i64, p, p, p]
    lib.oracle_rulebook_strided.restype = ctypes.c_int
    lib.oracle_rulebook_strided.argtypes = [p, i64, p, p, i64, p, p, p]
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def grid_size_of(voxel_size, point_cloud_range):
    """VoxelGenerator.__init__, voxel_generator.py:11-22 (float32 arithmetic, np.round)."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    v = np.asarray(voxel_size, dtype=np.float32)
    return np.round((r[3:] - r[:3]) / v).astype(np.int64)


def voxelize(points, voxel_size, point_cloud_range):
    """points_to_voxel(points, voxel_size, coors_range, reverse_index=True),
    voxel_generator.py:55-95.  Returns (coors int32[M,3] zyx, point_voxel_ids int32[N]).

    ``voxel_size`` / ``point_cloud_range`` are taken as float32 arrays, as
    VoxelGenerator stores them (voxel_generator.py:15-16); the per-point
    arithmetic then runs in numpy's promotion of (points.dtype, float32).
    """
    points = np.ascontiguousarray(points)
    assert points.ndim == 2 and points.shape[1] >= 3
    lib = _load()
    n, stride = points.shape
    if points.dtype == np.float32:
        fn, dt = lib.oracle_voxelize_f32, np.float32
    elif points.dtype == np.float64:
        fn, dt = lib.oracle_voxelize_f64, np.float64
    else:
        raise TypeError(points.dtype)
    # float32 -> float64 widening of the float32-stored constants is exact
    vs = np.asarray(voxel_size, dtype=np.float32).astype(dt)
    rng = np.asarray(point_cloud_range, dtype=np.float32).astype(dt)
    coors = np.zeros((max(n, 1), 3), dtype=np.int32)
    ids = np.full((n,), -1, dtype=np.int32)
    m = fn(_ptr(points), n, stride, _ptr(vs), _ptr(rng), _ptr(coors), _ptr(ids))
    assert m >= 0
    return coors[:m].copy(), ids


def ingroup_rank(group_inds):
    """get_inner_win_inds (ingroup_inds.py:7-20) with the canonical stable order."""
    g = np.ascontiguousarray(group_inds, dtype=np.int64)
    out = np.empty_like(g)
    rc = _load().oracle_ingroup_rank(_ptr(g), g.shape[0], _ptr(out))
    assert rc == 0
    return out


def rulebook_subm(coords, spatial_shape):
    """[27, M] int32 neighbour table of a 3x3x3 submanifold conv (a9)."""
    c = np.ascontiguousarray(coords, dtype=np.int32)
    shp = np.ascontiguousarray(spatial_shape, dtype=np.int32)
    m = c.shape[0]
    nbr = np.empty((27, m), dtype=np.int32)
    rc = _load().oracle_rulebook_subm(_ptr(c), m, _ptr(shp), _ptr(nbr))
    assert rc == 0
    return nbr


def downsample_coords(coords, spatial_shape):
    """Active output sites of SparseConv3d(k=3, s=2, p=1) in canonical order (a10)."""
    c = np.ascontiguousarray(coords, dtype=np.int32)
    shp = np.ascontiguousarray(spatial_shape, dtype=np.int32)
    m = c.shape[0]
    out = np.empty((8 * max(m, 1), 4), dtype=np.int32)
    shp_out = np.empty((3,), dtype=np.int32)
    mo = _load().oracle_downsample_coords(_ptr(c), m, _ptr(shp), _ptr(out), _ptr(shp_out))
    assert mo >= 0
    return out[:mo].copy(), shp_out


def rulebook_strided(coords_in, shape_in, coords_out, shape_out):
    """(nbr_fwd [27,M_out], nbr_inv [27,M_in]) for the strided conv and its inverse (a10, a11)."""
    ci = np.ascontiguousarray(coords_in, dtype=np.int32)
    co = np.ascontiguousarray(coords_out, dtype=np.int32)
    si = np.ascontiguousarray(shape_in, dtype=np.int32)
    so = np.ascontiguousarray(shape_out, dtype=np.int32)
    fwd = np.empty((27, co.shape[0]), dtype=np.int32)
    inv = np.empty((27, ci.shape[0]), dtype=np.int32)
    rc = _load().oracle_rulebook_strided(
        _ptr(ci), ci.shape[0], _ptr(si), _ptr(co), co.shape[0], _ptr(so), _ptr(fwd), _ptr(inv)
    )
    assert rc == 0
    return fwd, inv


def prepare_voxel_labels(point_voxel_ids, point_labels, n_voxels, ignore_index=255):
    """seg3d/datasets/waymo_dataset.py:213-246 restated with numpy: per voxel np.argmax of a 256-bin counter over its
    points' labels (first maximum = smallest label on ties), ignore_index for voxels without a point."""
    import numpy as np
    ids = np.asarray(point_voxel_ids).astype(np.int64)
    lab = np.asarray(point_labels).astype(np.int64)
    out = np.full((int(n_voxels),), ignore_index, dtype=np.uint8)
    ok = ids != -1
    if ok.any():
        counter = np.zeros((int(n_voxels), 256), dtype=np.int32)
        np.add.at(counter, (ids[ok], lab[ok]), 1)
        seen = counter.sum(1) > 0
        out[seen] = counter[seen].argmax(1).astype(np.uint8)
    return out

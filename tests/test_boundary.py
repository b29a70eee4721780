"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, the product package never reaches into oracle/, and the ops refuse to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from openseg3d_amd import _lib
    declared = _lib.header_symbols()
    assert len(declared) >= 25
    assert sorted(_lib.SIGNATURES) == declared, "binding table and include/seg3d_hip.h disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in seg3d_hip.h but not exported"
    assert _lib.load().seg3d_abi_version() == _lib.ABI_VERSION


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "seg3d_hip.h")).read()
    for cite in ("voxel_generator.py:55-153", "ingroup_inds_cuda.cu:12-25", "spconv_utils.py:13-32",
                 "swformer_utils.py", "cosine_msa.py:115-177", "voxel_pooling_cuda.cu:10-79", "voxel_to_point.py:4-17"):
        assert cite in text, cite


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "openseg3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                continue
            src = open(os.path.join(dirpath, f)).read()
            uses = re.findall(r"^\s*(?:from|import)\s+oracle\b.*$", src, flags=re.M)
            if f == "selfcheck.py":  # smoke()'s checker, allowed by the rules
                continue
            assert not uses, f"{f} imports the oracle: {uses}"
            assert "/root/reference" not in src, f"{f} reads the reference tree at run time"


def test_ops_fail_loudly_without_gpu():
    from openseg3d_amd import _lib, ops
    x = torch.zeros(4, 6)
    with pytest.raises(_lib.Seg3dError):
        ops.voxelize(x, [0.1, 0.1, 0.1], [-72, -72, -2, 72, 72, 4.4])
    with pytest.raises(_lib.Seg3dError):
        ops.voxel_to_point(torch.zeros(3, 4), torch.zeros(5, dtype=torch.long))
    with pytest.raises(_lib.Seg3dError):
        ops.get_inner_win_inds(torch.zeros(5, dtype=torch.long))


def test_host_only_entry_points():
    from openseg3d_amd import _lib, ops
    assert ops.grid_size([0.1, 0.1, 0.1], [-72, -72, -2, 72, 72, 4.4]) == [1440, 1440, 64]
    assert ops.grid_size([0.05, 0.012, 0.1], [0, -3.1415926, -2, 75.2, 3.1415926, 5.2]) == [1504, 524, 72]
    assert _lib.query("seg3d_voxelize_workspace_bytes", 180000) > 180000 * 12
    assert _lib.query("seg3d_coord_hash_bytes", 1000) == 2048 * 12
    # argument validation happens before anything is enqueued: a null call is rejected, not launched
    assert _lib.load().seg3d_spconv_fwd(None, None, 10, 10, None, 0, None, 48, 48, None, None, None) == _lib.EINVAL
    assert _lib.load().seg3d_spconv_fwd(None, None, 10, 10, None, 4, None, 50, 48, None, None, None) == _lib.EINVAL
    # the six-product Linear (csrc/linear_x6.hip) takes cin % 32 == 0 and cout % 64 == 0 and says so through its size query;
    # three planes of 8-element fragments per (32-channel step, 16-column tile)
    assert _lib.query("seg3d_linear_packed_bytes_x6", 128, 256) == (128 // 32) * (256 // 64) * 4 * 3 * 64 * 16 == 128 * 256 * 6
    assert _lib.query("seg3d_linear_packed_bytes_x6", 96, 256) == 96 * 256 * 6
    for cin, cout in ((48, 64), (64, 96), (6, 64), (64, 22), (0, 64)):
        assert _lib.query("seg3d_linear_packed_bytes_x6", cin, cout) == 0
    assert _lib.load().seg3d_linear_fwd_x6(None, 10, None, None, None, None, 0, 128, 256, None, None) == _lib.EINVAL
    assert _lib.load().seg3d_linear_fwd_x6(None, 0, None, None, None, None, 0, 128, 256, None, None) == 0  # nothing to do
    # both weight-gradient kernels fit the workspace the query reports; the opt-in switch validates its argument
    assert _lib.query("seg3d_linear_wgrad_workspace_bytes", 58453, 192, 192) >= 8 * (192 * 192 + 192) * 4
    assert _lib.load().seg3d_debug_set_wgrad_lds(2) == _lib.EINVAL and _lib.load().seg3d_debug_set_wgrad_lds(-1) == 0


@pytest.mark.parametrize("tag,rng,vs", [("cart", [-72, -72, -2, 72, 72, 4.4], [0.1, 0.1, 0.1]),
                                        ("cyl", [0, -3.1415926, -2, 75.2, 3.1415926, 5.2], [0.05, 0.012, 0.1])])
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_host_voxelizer_bit_exact_vs_reference(tag, rng, vs, dt):
    """VoxelGenerator.generate runs in DataLoader workers and TTA without a GPU context, so the CPU entry
    "must stay".  The library's own host entry (seg3d_voxelize_host_f32 / _f64, plain C++ in libseg3d_hip.so, no import of
    oracle/) against the outputs of the reference's points_to_voxel (tests/golden/voxelize.npz): coordinates in first-seen
    order and point -> voxel ids bit for bit, float32 and float64 rows, cartesian and cylinder grids."""
    import numpy as np
    from openseg3d_amd import batch, ops
    d = np.load(os.path.join(ROOT, "tests", "golden", "voxelize.npz"))
    k = f"{tag}_{dt}"
    pts = d[k + "_points"]
    gen = batch.VoxelGenerator(vs, rng)
    coors, ids = gen.generate(pts)  # numpy in -> the host entry, numpy out (voxel_generator.py:24-26)
    assert isinstance(coors, np.ndarray) and coors.dtype == np.int32 and ids.dtype == np.int32
    assert np.array_equal(coors, d[k + "_coors"]) and np.array_equal(ids, d[k + "_ids"])
    assert gen.grid_size.tolist() == d[tag + "_grid"].tolist()
    # collated form: a batch column in front, the same sample twice -> the second copy's voxels are new rows of batch 1
    rows = np.concatenate([np.pad(pts, ((0, 0), (1, 0)), constant_values=b) for b in (0.0, 1.0)])
    c4, i2 = ops.voxelize_host(rows, vs, rng, xyz_col=1, batch_col=0)
    m, n = coors.shape[0], pts.shape[0]
    assert np.array_equal(c4[:m, 1:], coors) and (c4[:m, 0] == 0).all() and np.array_equal(c4[m:, 1:], coors) and (c4[m:, 0] == 1).all()
    assert np.array_equal(i2[:n], ids) and np.array_equal(i2[n:], np.where(ids >= 0, ids + m, -1))


def test_host_voxelizer_edge_cases():
    import numpy as np
    from openseg3d_amd import _lib, ops
    vs, rng = [0.1, 0.1, 0.1], [-72, -72, -2, 72, 72, 4.4]
    c, i = ops.voxelize_host(np.zeros((0, 6), np.float32), vs, rng)
    assert c.shape == (0, 4) and i.shape == (0,)
    p = np.array([[0, 0, 0, 0, 0, 0], [1e6, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [0.05, 0.05, 0.05, 0, 0, 0],
                  [np.nan, 0, 0, 0, 0, 0], [72.0, 0, 0, 0, 0, 0], [-72.0, -72.0, -2.0, 0, 0, 0]], np.float32)
    c, i = ops.voxelize_host(p, vs, rng)
    assert i.tolist() == [0, -1, 0, 0, -1, -1, 1] and c.tolist() == [[0, 20, 720, 720], [0, 0, 0, 0]]
    with pytest.raises(_lib.Seg3dError):
        ops.voxelize_host(np.zeros((4, 6), np.int32), vs, rng)
    with pytest.raises(_lib.Seg3dError):  # a workspace that is too small is refused, not overrun
        ws = np.empty((64,), np.uint8)
        cnt = np.zeros((1,), np.int32)
        out_c, out_i = np.empty((7, 4), np.int32), np.empty((7,), np.int32)
        F3, F6 = ctypes.c_float * 3, ctypes.c_float * 6
        _lib.call("seg3d_voxelize_host_f32", p.ctypes.data, 7, 6, 0, -1, F3(*vs), F6(*rng), out_c.ctypes.data, out_i.ctypes.data,
                  cnt.ctypes.data, ws.ctypes.data, ws.size)

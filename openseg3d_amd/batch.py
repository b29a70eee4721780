"""GPU-side batch assembly: raw per-sample point rows -> the ``batch_dict`` Segformer.forward reads.

Replaces, for the bench / test path, VoxelGenerator.generate in DataLoader workers
(seg3d/core/voxel/voxel_generator.py:24-26, a serial numba loop over a 531 MB dense grid per sample)
plus WaymoDataset.collate_batch (waymo_dataset.py:339-376) and load_data_to_gpu (data_utils.py:6-15):
points are copied to the device once and the whole collated batch is voxelized there in one pass.
Dtypes follow load_data_to_gpu: everything float32 except point_voxel_ids (int64)."""
import numpy as np
import torch

from . import ops


def collate_points(samples, device):
    """list of [N_i, D] arrays -> float32 [sum N, 1+D] with the batch index in column 0."""
    rows = [np.concatenate([np.full((s.shape[0], 1), b, dtype=s.dtype), s], axis=1) for b, s in enumerate(samples)]
    pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(rows, axis=0)))
    return pts.to(device=device)


def batch_from_resident(points, row_offsets, voxel_size, point_cloud_range, image_features=None, cylinder=False):
    """points: collated [sum N, 1+D] tensor already in HBM; row_offsets: cumulative rows per sample (ints) --
    for multi-sweep batches the cumulative CURRENT-sweep rows (collate_batch's cur_point_count,
    waymo_dataset.py:367-373); image_features: optional [sum N_current, 28] tensor; cylinder: the rows are cartesian and
    DATASET.USE_CYLINDER is set -- cart2polar runs on the device in front of the voxelizer (waymo_dataset.py:270-275)."""
    if cylinder:
        points = ops.cart2polar(points, xyz_col=1)
    coords, ids = ops.voxelize(points, voxel_size, point_cloud_range, xyz_col=1, batch_col=0)
    extra = {} if image_features is None else {"point_image_features": image_features}
    return {
        **extra,
        "points": points if points.dtype == torch.float32 else points.float(),
        "voxel_coords": coords.float(),
        "point_voxel_ids": ids.long(),
        "point_id_offset": torch.tensor(row_offsets, dtype=torch.float32, device=points.device),
        "point_row_offsets": [int(o) for o in row_offsets],
        "point_voxel_index": ops.SegmentIndex(ids, coords.shape[0]),
        "batch_size": len(row_offsets),
    }


def make_batch(samples, voxel_size, point_cloud_range, device="cuda", cylinder=False):
    pts = collate_points(samples, device)
    offsets = np.cumsum([s.shape[0] for s in samples]).tolist()
    return batch_from_resident(pts, offsets, voxel_size, point_cloud_range, cylinder=cylinder)


class VoxelGenerator:
    """``seg3d.core.voxel.VoxelGenerator`` (voxel_generator.py:5-52) on the device voxelizer: same constructor, same
    ``voxel_size`` / ``point_cloud_range`` / ``grid_size`` attributes (float32 / float32 / int64 numpy arrays, the grid
    rounded in float32 exactly as voxel_generator.py:15-18), same ``generate(points)`` contract

        points [N, >= 3] float32 / float64  ->  (coors int32 [M, 3] as (z, y, x) in first-seen order,
                                                  point_voxel_ids int32 [N], -1 = out of range)

    with the subtract and the true divide done in the points' own dtype.  Two entries, as SURVEY 8(b)
    asks: a NUMPY array is voxelized ON THE HOST by the library's own CPU entry (seg3d_voxelize_host_f32 / _f64: the
    reference's serial first-seen loop over a hash instead of the 531 MB dense grid; no HIP call, so it runs in forked
    DataLoader workers -- waymo_dataset.py:275 -- and test_time_aug.py:33 exactly where the reference's numba kernel ran)
    and returns numpy; a CUDA tensor stays on the device (seg3d_voxelize_f32 / _f64) and returns tensors -- the fast path
    of the bench and of a TTA loop whose 36 re-voxelizations per frame then never leave the card.  Both are product code
    of libseg3d_hip.so; neither is a fallback for the other (a CUDA tensor is never sent to the host entry, a numpy
    array never to the device)."""

    def __init__(self, voxel_size, point_cloud_range, device="cuda"):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._grid_size = np.round(grid_size).astype(np.int64)
        self._device = device

    def generate(self, points):
        """Generate voxels given points (voxel_generator.py:24-26, 55-95)."""
        if isinstance(points, np.ndarray):
            if points.ndim != 2 or points.shape[1] < 3:
                raise ValueError("points must be [N, >= 3]")
            coords, ids = ops.voxelize_host(points, self._voxel_size.tolist(), self._point_cloud_range.tolist())
            return np.ascontiguousarray(coords[:, 1:]), ids
        pts = points
        if pts.dim() != 2 or pts.shape[1] < 3:
            raise ValueError("points must be [N, >= 3]")
        coords, ids = ops.voxelize(pts, self._voxel_size.tolist(), self._point_cloud_range.tolist())
        return coords[:, 1:].contiguous(), ids  # the device voxelizer carries a batch column: (b, z, y, x) -> (z, y, x)

    @property
    def voxel_size(self):
        """list[float]: Size of a single voxel."""
        return self._voxel_size

    @property
    def point_cloud_range(self):
        """list[float]: Range of point cloud."""
        return self._point_cloud_range

    @property
    def grid_size(self):
        """np.ndarray: The size of grids."""
        return self._grid_size

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self._voxel_size}, point_cloud_range="
                f"{self._point_cloud_range.tolist()}, grid_size={self._grid_size.tolist()})")

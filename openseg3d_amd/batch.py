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

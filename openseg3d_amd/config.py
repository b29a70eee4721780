"""Configuration: the reference's defaults (seg3d/utils/config.py:9-79) and its YAML merge rule
(:81-117), so configs/waymo_*.yaml load unchanged.  ``easydict`` is not a dependency: ``Cfg`` is a
dict with attribute access and the same strict key/type check."""
import copy

import numpy as np
import yaml


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _to_cfg(d):
    return Cfg({k: _to_cfg(v) if isinstance(v, dict) and not _is_leaf_dict(k) else v for k, v in d.items()})


def _is_leaf_dict(key):
    return key == "LOSSES"


def _levels(spec):
    return {str(i): {"max_tokens": t, "batching_range": [lo, hi]} for i, (t, lo, hi) in enumerate(spec)}


DEFAULTS = {
    "DATASET": {
        "USE_MULTI_SWEEPS": False, "MAX_NUM_SWEEPS": 5, "NUM_SWEEPS": 3, "USE_CYLINDER": False,
        "POINT_CLOUD_RANGE": [-72, -72, -2, 72, 72, 4.4], "VOXEL_SIZE": [0.1, 0.1, 0.1], "DIM_POINT": 6,
        "USE_IMAGE_FEATURE": False, "DIM_IMAGE_FEATURE": 28, "NUM_CLASSES": 22, "CLASS_NAMES": [],
        "CLASS_WEIGHT": [], "PALETTE": [], "IGNORE_INDEX": 255, "AUG_DATA": True,
        "AUG_ROT_RANGE": [-0.78539816, 0.78539816], "AUG_SCALE_RANGE": [0.95, 1.05], "AUG_TRANSLATE_STD": 0.5,
        "AUG_SAMPLE_RATIO": 0.95, "AUG_SAMPLE_RANGE": 50.0, "AUG_COLOR_DROP_RATIO": 0.5, "VISUALIZE": False,
    },
    "MODEL": {
        "SEGMENTOR": "segformer", "LOSSES": {"ohem_ce": 1.0, "lovasz": 1.0}, "OHEM_KEEP_RATIO": 0.3,
        "OHEM_KEEP_THRESH": 0.7, "AUX_LOSS_WEIGHT": 0.4,
        "BATCHING_INFO": [
            _levels([(16, 0, 16), (64, 16, 64), (256, 64, 256), (800, 256, 100000)]),
            _levels([(32, 0, 32), (128, 32, 128), (512, 128, 512), (800, 512, 100000)]),
            _levels([(64, 0, 64), (160, 64, 160), (384, 160, 384), (800, 384, 100000)]),
            _levels([(128, 0, 128), (256, 128, 256), (512, 256, 512), (800, 512, 100000)]),
        ],
        "WINDOW_SHAPE": [10, 10, 8], "DEPTHS": [3, 4, 8, 3], "DROP_PATH_RATE": 0.3,
    },
    "TRAIN": {"OPTIMIZER": "adamw", "LR": 0.001, "WEIGHT_DECAY": 0.01, "MOMENTUM": 0.9,
              "LR_SCHEDULER": "warmup_poly_lr"},
}


def default_cfg():
    return _to_cfg(copy.deepcopy(DEFAULTS))


def _merge(a, b, path=""):
    for k, v in a.items():
        if k not in b:
            raise KeyError(f"{path}{k} is not a valid config key")
        if isinstance(b[k], Cfg) and isinstance(v, dict):
            _merge(v, b[k], path + k + ".")
            continue
        if type(b[k]) is not type(v):
            if isinstance(b[k], np.ndarray):
                v = np.array(v, dtype=b[k].dtype)
            else:
                raise ValueError(f"Type mismatch ({type(b[k])} vs. {type(v)}) for config key: {path}{k}")
        b[k] = v


def cfg_from_file(filename, cfg=None):
    cfg = default_cfg() if cfg is None else cfg
    with open(filename) as f:
        _merge(yaml.load(f, Loader=yaml.FullLoader) or {}, cfg)
    return cfg


class DatasetSpec:
    """The attributes ``build_segmentor`` reads off the dataset (segformer.py:16-47, waymo_dataset.py:16-60)."""

    def __init__(self, cfg):
        from . import ops
        d = cfg.DATASET
        self.dim_point = d.DIM_POINT
        self.use_cylinder = d.USE_CYLINDER
        self.use_multi_sweeps = d.USE_MULTI_SWEEPS
        self.use_image_feature = d.USE_IMAGE_FEATURE
        self.dim_image_feature = d.DIM_IMAGE_FEATURE
        self.voxel_size = list(d.VOXEL_SIZE)
        self.point_cloud_range = list(d.POINT_CLOUD_RANGE)
        self.grid_size = np.array(ops.grid_size(self.voxel_size, self.point_cloud_range), dtype=np.int64)
        self.num_classes = d.NUM_CLASSES
        self.ignore_index = d.IGNORE_INDEX

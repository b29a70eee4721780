"""Reference default hyper-parameters (seg3d/utils/config.py:9-70) as plain data for the tests."""
import copy

import numpy as np

BATCHING_INFO = [
    {0: {'max_tokens': 16, 'batching_range': [0, 16]}, 1: {'max_tokens': 64, 'batching_range': [16, 64]},
     2: {'max_tokens': 256, 'batching_range': [64, 256]}, 3: {'max_tokens': 800, 'batching_range': [256, 100000]}},
    {0: {'max_tokens': 32, 'batching_range': [0, 32]}, 1: {'max_tokens': 128, 'batching_range': [32, 128]},
     2: {'max_tokens': 512, 'batching_range': [128, 512]}, 3: {'max_tokens': 800, 'batching_range': [512, 100000]}},
    {0: {'max_tokens': 64, 'batching_range': [0, 64]}, 1: {'max_tokens': 160, 'batching_range': [64, 160]},
     2: {'max_tokens': 384, 'batching_range': [160, 384]}, 3: {'max_tokens': 800, 'batching_range': [384, 100000]}},
    {0: {'max_tokens': 128, 'batching_range': [0, 128]}, 1: {'max_tokens': 256, 'batching_range': [128, 256]},
     2: {'max_tokens': 512, 'batching_range': [256, 512]}, 3: {'max_tokens': 800, 'batching_range': [512, 100000]}},
]
WINDOW_SHAPE = [10, 10, 8]
DEPTHS = [3, 4, 8, 3]
CART_RANGE = [-72, -72, -2, 72, 72, 4.4]
CART_VOXEL = [0.1, 0.1, 0.1]
CYL_RANGE = [0, -3.1415926, -2, 75.2, 3.1415926, 5.2]
CYL_VOXEL = [0.05, 0.012, 0.1]
GRID_CART = np.array([1440, 1440, 64])
GRID_CYL = np.array([1504, 524, 72])


def swformer_param_shapes(c, depth, prefix=""):
    shapes = {}
    for i in range(depth):
        p = f"{prefix}layers.{i}."
        shapes.update({
            p + "win_attn.self_attn.in_proj_weight": (3 * c, c), p + "win_attn.self_attn.in_proj_bias": (3 * c,),
            p + "win_attn.self_attn.out_proj.weight": (c, c), p + "win_attn.self_attn.out_proj.bias": (c,),
            p + "win_attn.self_attn.tau": (1, 1, 1),
            p + "norm1.weight": (c,), p + "norm1.bias": (c,), p + "norm2.weight": (c,), p + "norm2.bias": (c,),
            p + "mlp.fc1.weight": (2 * c, c), p + "mlp.fc1.bias": (2 * c,),
            p + "mlp.fc2.weight": (c, 2 * c), p + "mlp.fc2.bias": (c,),
        })
    return shapes


def segformer_key_shapes(keys, dim_point):
    """tests/golden/segformer_keys.json was written for dim_point=6; the cylinder config widens the
    first BN/Linear to 8 (segformer.py:16-18)."""
    k = copy.deepcopy(keys)
    if dim_point != 6:
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            k[f"point_encoder.0.{leaf}"] = [dim_point]
        k["point_encoder.1.weight"] = [64, dim_point]
    return k

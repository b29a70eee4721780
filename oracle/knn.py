"""Oracle restatement of knn_query -- TEST INFRASTRUCTURE ONLY.

seg3d/ops/knn_query/src/knn_query_cuda.cu:67-112 + wrapper knn_query.py:7-24: for every query the k nearest
candidates of the same batch segment, ascending squared distance, unfilled slots (1e10, segment start),
wrapper returns sqrt.  Both buffers are read with a stride of 3 floats whatever their second dimension is
(:97-99) -- reproduced, because DeepFusionBlock passes [N, 6] rows (deep_fusion.py:31).
Canonical tie rule of this build: equal distances by ascending candidate index (stable sort).
float32 arithmetic in the kernel's order, no FMA: d2 = (dx*dx + dy*dy) + dz*dz.
"""
import numpy as np
import torch


def knn_query(nsample, xyz, new_xyz, offset, new_offset, chunk=2048):
    if new_xyz is None:
        new_xyz = xyz
    n, m = xyz.shape[0], new_xyz.shape[0]
    x = np.ascontiguousarray(xyz.detach().numpy(), dtype=np.float32).reshape(-1)[: 3 * n].reshape(n, 3)
    q = np.ascontiguousarray(new_xyz.detach().numpy(), dtype=np.float32).reshape(-1)[: 3 * m].reshape(m, 3)
    off = np.asarray(offset, dtype=np.int64).reshape(-1)
    noff = np.asarray(new_offset, dtype=np.int64).reshape(-1)
    idx = np.zeros((m, nsample), np.int32)
    d2o = np.full((m, nsample), np.float32(1e10), np.float32)
    for b in range(off.shape[0]):
        s, e = (0 if b == 0 else int(off[b - 1])), int(off[b])
        qs, qe = (0 if b == 0 else int(noff[b - 1])), (int(noff[b]) if b < off.shape[0] - 1 else m)
        qe = min(qe, m)
        idx[qs:qe] = s
        cand = x[s:e]
        if e <= s:
            continue
        for c0 in range(qs, qe, chunk):
            c1 = min(c0 + chunk, qe)
            d = q[c0:c1, None, :] - cand[None, :, :]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            order = np.argsort(d2, axis=1, kind="stable")[:, :nsample]
            kk = order.shape[1]
            idx[c0:c1, :kk] = (order + s).astype(np.int32)
            d2o[c0:c1, :kk] = np.take_along_axis(d2, order, axis=1)
    return torch.from_numpy(idx), torch.sqrt(torch.from_numpy(d2o))

"""CPU oracle for the OpenSeg3D sparse-voxel hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (plain C for the integer/index work, numpy /
torch-CPU for the floating-point work), the algorithms of the reference path
named in BASELINE.json (voxelize -> sparse conv -> sparse window attention ->
per-point logits).  Every function cites the reference file:line it follows.

Rules (enforced by tests/test_boundary.py):
  * nothing under ``openseg3d_amd/`` may import, call, link or execute anything
    from here -- the product path fails loudly when the HIP library is missing;
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg use it, and only as the checker / reported baseline.

Pinning status (see DESIGN.md "Oracle"):
  * voxelizer, window partition, positional embedding, masks, cosine window
    attention, SWFormer block, Segformer wiring: pinned against outputs of the
    reference's own Python, imported file-by-file in the build container by
    ``tests/golden/make_golden.py`` (fixtures committed under tests/golden/).
  * sparse convolution (third-party ``spconv``, not vendored, version unpinned
    -- requirements.txt:4) and ``torch_scatter``: **parity unpinned** against
    the third-party packages themselves; the restatement follows the reference
    call sites and is cross-checked against dense ``torch.nn.functional.conv3d``
    / ``index_reduce`` on densified grids (tests/test_oracle_sparse_conv.py).
"""
from .index_ops import (  # noqa: F401
    voxelize,
    ingroup_rank,
    rulebook_subm,
    downsample_coords,
    rulebook_strided,
)

/*
 * seg3d_hip.h -- C ABI of libseg3d_hip.so, the MI355X (gfx950) implementation of the
 * OpenSeg3D sparse-voxel segmentation hot path.
 *
 * Boundary: the reference exposes this path as Python callables
 * backed by pybind11/CUDA extension modules and by the third-party spconv /
 * torch_scatter packages.  This header is what a binding for that path binds instead:
 * plain pointers and sizes, explicit stream, no torch types, no allocation inside
 * (callers pass workspaces sized by the *_workspace_bytes queries), no exceptions.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter comment says "host";
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, so every entry point is hipGraph-capturable;
 *   - return value: 0 = ok, <0 = SEG3D_E* below (checked before anything is enqueued);
 *   - row-major, float32 features, int32 indices; -1 marks "none";
 *   - coordinates are rows [batch, z, y, x] (spconv order), spatial shapes are (z, y, x);
 *   - kernel offset k = (kz*3 + ky)*3 + kx, neighbour site = site + (kz-1, ky-1, kx-1).
 *
 * Each entry cites the reference interface (path:line under /root/reference) it replaces.
 */
#ifndef SEG3D_HIP_H
#define SEG3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEG3D_OK 0
#define SEG3D_EINVAL (-1)     /* bad argument (null pointer, unsupported size, ...) */
#define SEG3D_EWORKSPACE (-2) /* workspace smaller than the *_workspace_bytes query */
#define SEG3D_ELAUNCH (-3)    /* hipGetLastError() after a launch */

#define SEG3D_REDUCE_SUM 0
#define SEG3D_REDUCE_MEAN 1
#define SEG3D_REDUCE_MAX 2

/* ABI version; bumped whenever a signature below changes. */
int seg3d_abi_version(void);

/* Text of the HIP runtime error behind the calling thread's most recent SEG3D_ELAUNCH
 * ("<hipGetErrorString> (<hipGetErrorName>) at <file>:<line>"), "" if none yet.  The reference
 * surfaces argument failures as Python exceptions through TORCH_CHECK
 * (seg3d/ops/ingroup_inds/src/ingroup_inds.cpp:6-12) and checks nothing after its kernel launches
 * (ingroup_inds_cuda.cu:35-49); a C ABI that returns codes needs an accessor for the runtime's text.
 * The pointer stays valid for the life of the thread. */
const char* seg3d_last_error(void);

/* ------------------------------------------------------------------------------------------
 * a2  VoxelGenerator.__init__  -- seg3d/core/voxel/voxel_generator.py:11-22
 * grid = round((hi - lo) / voxel_size) evaluated in float32 (host helper, no GPU work).
 * voxel_size: host float[3] (x,y,z); range: host float[6] (xyz min, xyz max); grid_xyz: host int32[3].
 */
int seg3d_grid_size(const float* voxel_size, const float* range, int32_t* grid_xyz);

/* ------------------------------------------------------------------------------------------
 * a1 + a4  points_to_voxel / _points_to_voxel_reverse_kernel -- voxel_generator.py:55-153,
 *          batch id column + cumulative voxel-id offset of WaymoDataset.collate_batch --
 *          seg3d/datasets/waymo_dataset.py:339-365.
 * Hard voxelisation with first-seen voxel order, on device, for a whole collated batch:
 * c_j = floor((p_j - lo_j) / vs_j) with an IEEE subtract and true divide in the dtype of
 * `points`; a point is rejected (id -1) if any c_j is outside [0, grid_j).
 *   points      [n, row_stride] row-major; xyz at columns xyz_col..xyz_col+2;
 *               batch id (stored as a float, as the reference collates it) at batch_col, or
 *               batch_col < 0 for a single sample (batch id 0)
 *   voxel_size, range: host float[3], float[6]  (float32 constants, widened for the f64 entry)
 *   voxel_coords      out [>= n, 4]  rows [b, z, y, x] in first-seen order
 *   point_voxel_ids   out [n]        batch-global voxel row of each point, -1 if rejected
 *   n_voxels          out device int32[1]
 */
size_t seg3d_voxelize_workspace_bytes(int64_t n_points);
int seg3d_voxelize_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                       int32_t batch_col, const float* voxel_size, const float* range,
                       int32_t* voxel_coords, int32_t* point_voxel_ids, int32_t* n_voxels,
                       void* workspace, size_t workspace_bytes, void* stream);
int seg3d_voxelize_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                       int32_t batch_col, const float* voxel_size, const float* range,
                       int32_t* voxel_coords, int32_t* point_voxel_ids, int32_t* n_voxels,
                       void* workspace, size_t workspace_bytes, void* stream);

/* (b) The CPU entry of VoxelGenerator.generate -- seg3d/core/voxel/voxel_generator.py:24-26 (-> points_to_voxel :55-95,
 * _points_to_voxel_reverse_kernel :98-153), called by DataLoader workers (seg3d/datasets/waymo_dataset.py:275) and
 * test-time augmentation (test_time_aug.py:33): forked processes WITHOUT a GPU context, which SURVEY 8(b) says must keep
 * a CPU voxelizer.  Same contract as seg3d_voxelize_f32 / _f64 with every pointer a HOST pointer and no stream: the
 * reference's serial first-seen loop, the dense 531 MB lookup grid replaced by an open-addressing table in the caller's
 * workspace.  Plain host code: makes no HIP call (safe after fork, runs on a machine without a GPU).  n_voxels: host
 * int32[1]; voxel_coords host [>= n_points, 4] rows (b, z, y, x). */
size_t seg3d_voxelize_host_workspace_bytes(int64_t n_points);
int seg3d_voxelize_host_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                            int32_t batch_col, const float* voxel_size, const float* range,
                            int32_t* voxel_coords, int32_t* point_voxel_ids, int32_t* n_voxels,
                            void* workspace, size_t workspace_bytes);
int seg3d_voxelize_host_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                            int32_t batch_col, const float* voxel_size, const float* range,
                            int32_t* voxel_coords, int32_t* point_voxel_ids, int32_t* n_voxels,
                            void* workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------
 * a3  cart2polar -- seg3d/utils/pointops_utils.py:8-11, wired into the cylinder configs at
 *     seg3d/datasets/waymo_dataset.py:270-273: rows [.., x, y, z, f..] -> [.., rho, phi, z, x, y, f..]
 *     (row_stride + 2 columns; the `xyz_col` columns in front -- the batch index of a collated tensor -- are
 *     copied).  rho = sqrt(x*x + y*y) in the point dtype, bit-identical to numpy; phi = atan2(y, x): float32 rows
 *     get the double-precision result rounded once, float64 rows the device library's atan2 (numpy takes the host
 *     libm's atan2f / atan2, documented at 1 ulp: that column of the reference is machine-dependent in its last bit).
 */
int seg3d_cart2polar_f32(const float* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                         float* out /* [n_points, row_stride + 2] */, void* stream);
int seg3d_cart2polar_f64(const double* points, int64_t n_points, int32_t row_stride, int32_t xyz_col,
                         double* out /* [n_points, row_stride + 2] */, void* stream);

/* ------------------------------------------------------------------------------------------
 * a14  ingroup_inds_ext.forward(group_inds, out_inds) -- seg3d/ops/ingroup_inds/src/
 *      ingroup_inds.cpp:28-48, ingroup_inds_cuda.cu:12-25 (+ the CSR the callers rebuild from it).
 * Rank of every element inside its group.  The reference hands ranks out in atomic arrival
 * order; this build's canonical order is ascending element index (deterministic).
 *   group_ids [n] in [0, n_groups), or -1 = not in any group
 *   rank      out [n] or NULL   (-1 for skipped elements)
 *   order     out [n] or NULL   element indices grouped by group id, ascending inside a group
 *   offsets   out [n_groups+1] or NULL   CSR offsets into `order`
 */
size_t seg3d_group_index_workspace_bytes(int64_t n, int64_t n_groups);
int seg3d_group_index(const int32_t* group_ids, int64_t n, int64_t n_groups, int32_t* rank,
                      int32_t* order, int32_t* offsets, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * a8-a11  spconv indice generation (third-party spconv, call sites seg3d/utils/spconv_utils.py:13-32,
 *         seg3d/models/backbones/pointtransformer.py:26-34,73-81,159-166,184-189).
 * Coordinate hash (open addressing, 64-bit linear site key -> row) and the neighbour tables
 * ("rulebooks") built from it.  Tables are offset-major int32 [27][m_rows], -1 = inactive.
 */
size_t seg3d_coord_hash_bytes(int64_t m);
int seg3d_coord_hash_build(const int32_t* coords, int64_t m, const int32_t* shape_zyx /*host[3]*/,
                           void* table, size_t table_bytes, void* stream);
/* SubMConv3d(k=3, padding=1): nbr[k][i] = row of the active site at coords[i] + offset(k). */
int seg3d_rulebook_subm(const int32_t* coords, int64_t m, const int32_t* shape_zyx /*host[3]*/,
                        const void* table, size_t table_bytes, int32_t* nbr, void* stream);
/* SparseConv3d(k=3, stride=2, padding=1) output sites, ascending (b,z,y,x).
 * shape_out = floor((shape_in + 2 - 3) / 2) + 1.  coords_out must hold cap_out >= min(8*m_in, cells) rows. */
size_t seg3d_downsample_workspace_bytes(int32_t batch_size, const int32_t* shape_in_zyx /*host[3]*/);
int seg3d_downsample_coords(const int32_t* coords_in, int64_t m_in,
                            const int32_t* m_in_dev /* or NULL: exact row count on the DEVICE when m_in is only an upper
                               bound -- lets the three strided levels be chained without reading a count back */,
                            int32_t batch_size, const int32_t* shape_in_zyx /*host[3]*/, int32_t* coords_out,
                            int64_t cap_out, int32_t* m_out /*device*/, void* workspace, size_t workspace_bytes,
                            void* stream);
/* nbr_fwd[k][o] = fine row at 2*coords_out[o] + k - 1; nbr_inv[k][i] = coarse row o with
 * 2*o + k - 1 == coords_in[i] (the table SparseInverseConv3d reuses under the same indice_key). */
int seg3d_rulebook_strided(const int32_t* coords_out, int64_t m_out, int64_t m_in,
                           const int32_t* shape_in_zyx /*host[3]*/, const void* table_in,
                           size_t table_bytes, int32_t* nbr_fwd, int32_t* nbr_inv, void* stream);

/* ------------------------------------------------------------------------------------------
 * a9-a11  spconv SubMConv3d / SparseConv3d / SparseInverseConv3d forward + backward
 *         (same call sites as above).  One output-stationary gather-GEMM kernel serves all three:
 *             y[r] = bias + sum_k x[nbr[k][r]] . W_k
 *   weight      [cout, 27, cin]  (= [Cout,3,3,3,Cin], the layout the modules keep)
 *   w_packed    seg3d_spconv_packed_bytes(cin, cout, flags) bytes written by seg3d_spconv_pack_weight;
 *               cin/cout there are those of `weight`; flags bit0: operand is W_k^T (dgrad), bit1: offsets
 *               reversed (k -> 26-k), bit2: split-bf16 pack -- every fp32 weight stored as bf16 hi + bf16 lo and
 *               the product evaluated as 3 bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate, ~2^-16
 *               relative; 16/3 the fp32-MFMA rate).  Without bit2 the exact-fp32 MFMA path is used.
 *               seg3d_spconv_fwd must be given the same flags the pack was made with.
 * dgrad: seg3d_spconv_fwd over the transposed pair list with a W^T pack and cin/cout swapped:
 *        subm: same table, flags=3;  strided conv: the inverse table, flags=1;  inverse conv: the
 *        forward table, flags=1.
 * wgrad: dw[co][k][ci] = sum_r x[nbr[k][r]][ci] * dy[r][co].
 * cin and cout must be multiples of 16 (cin multiple of 4 for fwd).
 */
size_t seg3d_spconv_packed_bytes(int32_t cin, int32_t cout, int32_t flags);
int seg3d_spconv_pack_weight(const float* weight, int32_t cin, int32_t cout, int32_t flags,
                             void* w_packed, void* stream);
int seg3d_spconv_fwd(const float* x, const int32_t* nbr, int64_t m_out, int64_t m_in,
                     const void* w_packed, int32_t pack_flags, const float* bias /*or NULL*/,
                     int32_t cin, int32_t cout, float* y,
                     const int32_t* row_order /* [m_out] permutation: processing order of the output rows, or NULL.
                        Results do not depend on it; strided / inverse tables run ~2x faster when rows are grouped by
                        coordinate parity (same active offsets per tile).  Ignored by the exact-fp32 path. */,
                     void* stream);
/* Inference form of a conv block -- conv -> BatchNorm1d(eval) -> (+ residual) -> ReLU (ConvModule spconv_utils.py:13-32,
 * SparseBasicBlock pointtransformer.py:47-66 / spconv_unet.py:46-66): the caller folds the BatchNorm affine into the
 * weights before packing (W' = W * gamma * rstd per output channel) and into the bias (beta - mean * gamma * rstd
 * (+ conv bias * gamma * rstd)); the block is then ONE launch, y = act(conv_W'(x) + bias (+ addend [m_out, cout])).
 * Split-bf16 packs (flags bit2) only. */
int seg3d_spconv_fwd_act(const float* x, const int32_t* nbr, int64_t m_out, int64_t m_in, const void* w_packed,
                         int32_t pack_flags, const float* bias /*or NULL*/, const float* addend /*or NULL*/,
                         int32_t relu, int32_t cin, int32_t cout, float* y, const int32_t* row_order /*or NULL*/,
                         void* stream);
/* The same block with the sparse-conv feature maps STORED in bf16 (opt-in storage mode, BASELINE configs[4]; the
 * reference's only reduced-precision hook is seg3d/ops/voxel_pooling/voxel_pooling.py:12): x holds float32 (x_bf16 = 0)
 * or bf16 (x_bf16 = 1) rows, y and the residual addend are bf16 [m_out, cout]; accumulation, bias and activation stay
 * float32, the weights stay split-bf16 hi + lo.  A bf16 row is its own hi part: its products take two MFMAs instead of
 * three and its gather moves half the bytes.  Inference only (no backward entry point takes bf16 rows). */
int seg3d_spconv_fwd_act_bf16(const void* x, int32_t x_bf16, const int32_t* nbr, int64_t m_out, int64_t m_in,
                              const void* w_packed, int32_t pack_flags, const float* bias,
                              const void* addend_bf16, int32_t relu, int32_t cin, int32_t cout, void* y_bf16,
                              const int32_t* row_order, void* stream);
/* Wide layers (cin >= 192): the same contract as seg3d_spconv_fwd_act with the operand conversion hoisted out of the
 * gather-GEMM.  seg3d_spconv_presplit writes x [m_in, cin] once as bf16 hi plane | lo plane ([m_in + 1, cin] each; the
 * extra row is zero and stands for every inactive table entry) into xs (seg3d_spconv_presplit_bytes);
 * seg3d_spconv_fwd_presplit then moves rows and weights global -> LDS by LDS-DMA and multiplies: no operand passes
 * through a vector register before it is an MFMA fragment.  cin % 32 == 0, cout % 96 == 0, split-bf16 packs only.
 * Same call sites as seg3d_spconv_fwd (spconv_utils.py:13-32, pointtransformer.py:69-113, 159-166). */
size_t seg3d_spconv_presplit_bytes(int64_t m_in, int32_t cin);
int seg3d_spconv_presplit(const float* x, int64_t m_in, int32_t cin, void* xs, void* stream);
int seg3d_spconv_fwd_presplit(const void* xs, const int32_t* nbr, int64_t m_out, int64_t m_in, const void* w_packed,
                              int32_t pack_flags, const float* bias /*or NULL*/, const float* addend /*or NULL*/,
                              int32_t relu, int32_t cin, int32_t cout, float* y,
                              const int32_t* row_order /*or NULL*/, void* stream);
/* a9 "row image" schedule of the same gather-GEMM for submanifold tables (spconv.SubMConv3d, call sites
 * seg3d/utils/spconv_utils.py:15-17, seg3d/models/backbones/pointtransformer.py:26-34,47-66,88-113; spconv builds the
 * equivalent "indice pairs" once per indice_key and reuses them in every layer that names the key).  A TILE PLAN is built
 * once per neighbour table (site level): output rows in Morton order of their coordinates, 128 per tile, inside a tile
 * sorted by neighbour mask; per tile the list of DISTINCT input rows (<= 512, else the tile runs offset by offset) and,
 * per (offset, row), the LDS slot of the neighbour's row.  seg3d_spconv_fwd_tiled then loads and splits every distinct
 * input row of a tile ONCE per 32-channel slice into an LDS image and feeds all 27 offsets from it (spconv_split gathers and
 * splits per (row, offset) pair: 6.5 - 17 times per row).  Results are bit-identical to seg3d_spconv_fwd_act on the same
 * operands (same products in the same order per output row) for cout a multiple of 96 or 128.  EXCEPTION: cout 32 / 48
 * (and therefore cin 32 / 48 of the input-gradient conv, whose cout is the forward's cin) run the offset-split layouts --
 * the four waves of a workgroup take a quarter of a tile's 27 offsets each and their partial tiles are summed through LDS
 * in a fixed order: deterministic run to run, but ANOTHER summation order than seg3d_spconv_fwd_act's ascending offsets,
 * so the two entry points agree there to fp32 round-off (a few 1e-7 relative), not bit for bit.  coords [m_out, 4] (b, z, y, x) int32 are the sites the
 * table's rows belong to; nbr [27][m_out] may gather from any row set (m_in rows).  cout a multiple of 96 or 128, or 32 / 48
 * (seg3d_spconv_tiled_supported), split-bf16 packs only; forward and dgrad (W^T pack, flipped offsets) share one plan. */
size_t seg3d_conv_plan_bytes(int64_t m_out);
size_t seg3d_conv_plan_workspace_bytes(int64_t m_out);
int seg3d_conv_plan_build(const int32_t* coords, const int32_t* nbr, int64_t m_out, void* plan, void* workspace,
                          size_t workspace_bytes, void* stream);
int32_t seg3d_spconv_tiled_supported(int32_t cin, int32_t cout);
int seg3d_spconv_fwd_tiled(const float* x, const int32_t* nbr, const void* plan, int64_t m_out, int64_t m_in,
                           const void* w_packed, int32_t pack_flags, const float* bias /*or NULL*/,
                           const float* addend /*or NULL*/, int32_t relu, int32_t cin, int32_t cout, float* y, void* stream);
/* The tiled schedule with the feature maps STORED in bf16 -- seg3d_spconv_fwd_act_bf16's contract (opt-in storage mode,
 * BASELINE configs[4]): x float32 (x_bf16 = 0) or bf16 (x_bf16 = 1) rows, y and the residual addend bf16 [m_out, cout].  bf16
 * rows stage only a hi image (half the bytes) and take two MFMAs per product. */
int seg3d_spconv_fwd_tiled_bf16(const void* x, int32_t x_bf16, const int32_t* nbr, const void* plan, int64_t m_out,
                                int64_t m_in, const void* w_packed, int32_t pack_flags, const float* bias /*or NULL*/,
                                const void* addend_bf16 /*or NULL*/, int32_t relu, int32_t cin, int32_t cout, void* y_bf16,
                                void* stream);
/* Test hook: force the column-block width (x16 columns) of the split-bf16 gather-GEMM so that every kernel
 * instantiation can be pinned against the oracle at any row count (0 = automatic choice; 1, 2, 3, 4, 6, 12).
 * Same effect as the SEG3D_CONV_NBT environment variable, which is read once when the library loads.
 * The reference has no counterpart (spconv chooses its own tiles). */
int seg3d_debug_set_conv_nbt(int32_t nbt);
/* Test aid for the opt-in row-streaming schedule of the dense Linear layers with cin <= 192 (csrc/linear_stream.hip, the
 * nn.Linear calls of point_transformer_layer.py:260-298 / cosine_msa.py:58-63,403; bit-identical to the default schedule,
 * measured slower, SEG3D_LINEAR_STREAM=1): 1 = on, 0 = off, -1 = the environment's choice. */
int seg3d_debug_set_linear_stream(int32_t on);
size_t seg3d_spconv_wgrad_workspace_bytes(int64_t m_out, int32_t cin, int32_t cout);
int seg3d_spconv_wgrad(const float* x, const float* dy, const int32_t* nbr, int64_t m_out,
                       int64_t m_in, int32_t cin, int32_t cout, int32_t flags /* bit2: split-bf16 */,
                       float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* The first half of seg3d_spconv_wgrad alone (split-bf16 arithmetic): partial blocks part[chunks][27 * cin * cout] stay in the
 * workspace, *chunks (host) = their count (0 for m_out == 0: the sum then writes zeros); the fixed-order sum is queued by the
 * caller with the other parameter-gradient sums of the backward pass (seg3d_reduce_partials_batched).  spconv has no
 * counterpart: it accumulates weight gradients with atomics inside its own kernels (call sites spconv_utils.py:13-32). */
int seg3d_spconv_wgrad_partials(const float* x, const float* dy, const int32_t* nbr, int64_t m_out, int64_t m_in,
                                int32_t cin, int32_t cout, void* workspace, size_t workspace_bytes, int32_t* chunks,
                                void* stream);
/* The same with the layer's input rows stored as bf16 [m_in, cin] -- BASELINE configs[4] names bf16, the reference has no
 * reduced-precision mode, so the mode is build-defined and OPT-IN (SEG3D_TRAIN_STORAGE=bf16): the sparse-conv /
 * Linear autograd functions (call sites spconv_utils.py:13-32, point_transformer_layer.py:260-298) keep a bf16 COPY of their
 * input for the backward pass instead of the fp32 tensor; every tensor of the graph, and therefore every gradient, stays
 * fp32.  A bf16 row is its own high half: 8-byte gathers, no split of x, two MFMAs per product; dy stays fp32 / split. */
int seg3d_spconv_wgrad_partials_xbf16(const uint16_t* x_bf16, const float* dy, const int32_t* nbr, int64_t m_out,
                                      int64_t m_in, int32_t cin, int32_t cout, void* workspace, size_t workspace_bytes,
                                      int32_t* chunks, void* stream);

/* a6, a22  weight gradient of the dense per-point / per-voxel Linear layers (segformer.py:21-32,58-76,
 * point_transformer_layer.py:260-276, cosine_msa.py:58-63,403):  dw[cout][cin] = dy^T . x  over m rows.
 * Split-bf16 tall-skinny GEMM (rows = MFMA K dimension), workgroup-tiled; partial blocks per row chunk go to
 * the workspace and are summed in a fixed order, so dw/db are bit-reproducible and need no memset.
 * db (nullable) receives the bias gradient = column sums of dy.  cin, cout multiples of 4. */
size_t seg3d_linear_wgrad_workspace_bytes(int64_t m, int32_t cin, int32_t cout);
int seg3d_linear_wgrad(const float* x, const float* dy, int64_t m, int32_t cin, int32_t cout, float* dw,
                       float* db /* [cout] or NULL */, void* workspace, size_t workspace_bytes, void* stream);
/* The two halves of seg3d_linear_wgrad apart (same kernels): seg3d_linear_wgrad_partials leaves the per-chunk partial
 * blocks part[chunks][cin*cout + cout] in the workspace and reports the chunk count through *chunks (host memory; 0 when
 * m == 0); seg3d_reduce_partials sums part[chunks][n] in a fixed order into dw[0, nw) and db[0, n - nw) (db nullable; n, nw
 * multiples of 4; chunks == 0 writes zeros).  seg3d_reduce_partials_batched runs many such sums in ONE launch: jobs =
 * device array of { const float* part; float* dw; float* db; int64_t n, nw; int32_t chunks, reserved; int64_t first_block }
 * (56 bytes) sorted by first_block, a job owning ceil(n / 256) blocks; total_blocks = their sum.  A training step's
 * ~160 parameter-gradient sums (Linear weights / biases, LayerNorm gamma / beta) become one launch at the end of the
 * backward pass (openseg3d_amd/ops.py, deferred join).  The reference has no counterpart: torch.autograd sums inside each
 * layer's own GEMM / reduction kernels (point_transformer_layer.py:260-298). */
int seg3d_linear_wgrad_partials(const float* x, const float* dy, int64_t m, int32_t cin, int32_t cout, int32_t with_bias,
                                void* workspace, size_t workspace_bytes, int32_t* chunks, void* stream);
/* ... with x stored as bf16 [m, cin] (the opt-in training copies, see seg3d_spconv_wgrad_partials_xbf16) */
int seg3d_linear_wgrad_partials_xbf16(const uint16_t* x_bf16, const float* dy, int64_t m, int32_t cin, int32_t cout,
                                      int32_t with_bias, void* workspace, size_t workspace_bytes, int32_t* chunks,
                                      void* stream);
/* A/B switch of the LDS-shared dense weight-gradient kernel (openseg3d_amd/csrc/wgrad_dense_lds.hip; OFF by default, it did
 * not win): 1 = on, 0 = off, -1 = SEG3D_WGRAD_LDS's choice.  Takes cout % 192 == 0, cin % 96 == 0, m >= 4096; the workspace
 * query covers both kernels. */
int seg3d_debug_set_wgrad_lds(int32_t on);
int seg3d_reduce_partials(const float* part, int32_t chunks, int64_t n, int64_t nw, float* dw, float* db, void* stream);
int seg3d_reduce_partials_batched(const void* jobs, int32_t n_jobs, int64_t total_blocks, void* stream);
/* a6  exact-fp32 variant for the per-point MLPs (segformer.py:21-32,58-76), whose split-bf16 forward error would land
 * directly on the logits: the same contract as seg3d_linear_* on v_mfma_f32_16x16x4_f32;
 * cin, cout multiples of 16.  Forward: transpose = 0; input gradient: transpose = 1 with cin/cout swapped. */
size_t seg3d_linear_packed_bytes_f32(int32_t cin, int32_t cout);
int seg3d_linear_pack_weight_f32(const float* weight, int32_t cin, int32_t cout, int32_t transpose,
                                 void* w_packed, void* stream);
int seg3d_linear_fwd_f32(const float* x, int64_t m, const void* w_packed, const float* bias /*or NULL*/,
                         int32_t cin, int32_t cout, float* y, void* stream);
/* a6  fp32-GRADE variant on the bf16 matrix pipe (round 5; the default of the per-point MLPs, segformer.py:21-32,58-76):
 * every operand split three ways (x = h + m + l, bf16 each: exact), six v_mfma_f32_16x16x32_bf16 per product keep every
 * term down to 2^-16 of the leading one, the dropped ones are <= 2^-24 -- the error of an fp32 sum, at 6/16 of the fp32
 * MFMA's time (openseg3d_amd/csrc/linear_x6.hip).  cin a multiple of 32, cout of 64 (packed_bytes = 0 otherwise: the
 * caller takes seg3d_linear_fwd_f32).  Optional epilogue y * scale[col] + shift[col] (both or neither) and ReLU: the eval
 * form of the BatchNorm1d + ReLU behind these layers (seg3d/models/segmentors/segformer.py:21-32). */
size_t seg3d_linear_packed_bytes_x6(int32_t cin, int32_t cout);
int seg3d_linear_pack_weight_x6(const float* weight, int32_t cin, int32_t cout, int32_t transpose, void* w_packed,
                                void* stream);
int seg3d_linear_fwd_x6(const float* x, int64_t m, const void* w_packed, const float* bias /*or NULL*/,
                        const float* scale /*or NULL*/, const float* shift /*or NULL*/, int32_t relu, int32_t cin,
                        int32_t cout, float* y, void* stream);

/* Batched split-bf16 packing: one launch for every conv / Linear weight whose pack is stale (in training all of
 * them, twice: W for forward, W^T for the input gradient).  `jobs` is a DEVICE array of n_jobs records sorted by
 * first_block; job i covers blocks [first_block_i, first_block_{i+1}) of 256 work items each, one item = 16 packed
 * elements (ceil(packed_bytes / 32 / 256) blocks); dst as for seg3d_spconv_pack_weight (flags bit2 set) /
 * seg3d_linear_pack_weight.  kk = 27 (conv, weight [Cout,27,Cin]) or 1 (Linear, weight [Cout,Cin]). */
typedef struct {
  const float* src;     /* weight in the reference layout */
  void* dst;            /* packed stream */
  int32_t cin, cout;    /* of the weight as stored */
  int32_t kk, transpose, flip, reserved;
  int64_t first_block;
} seg3d_pack_job;       /* 48 bytes */
int seg3d_pack_weights_batched(const void* jobs, int32_t n_jobs, int64_t total_blocks, void* stream);

/* a6, a22  forward / input gradient of the same layers: y[m, cout] = x[m, cin] . W^T + bias, as the
 * single-offset case of the split-bf16 gather-GEMM kernel (W fragments staged through LDS once per
 * 128-row tile).  weight is torch's [cout, cin]; transpose=1 packs W itself as the operand, i.e.
 * dx[m, cin] = dy[m, cout] . W is seg3d_linear_fwd(dy, ..., cin_of_call = cout, cout_of_call = cin).
 * Operand rows (cin of the call) must be a multiple of 8, columns (cout of the call) a multiple of 16. */
size_t seg3d_linear_packed_bytes(int32_t cin, int32_t cout, int32_t transpose);
int seg3d_linear_pack_weight(const float* weight, int32_t cin, int32_t cout, int32_t transpose,
                             void* w_packed, void* stream);
int seg3d_linear_fwd(const float* x, int64_t m, const void* w_packed, const float* bias /*or NULL*/,
                     const float* addend /* [m, cout] added in the epilogue, or NULL */, int32_t cin, int32_t cout,
                     float* y, void* stream);
/* y = (x + x_add) W^T + b without writing x + x_add: the q | k half of the cosine attention's in-projection reads
 * x + pos (cosine_msa.py:58-63 via point_transformer_layer.py:289-291).  Inference path; same kernel, the sum is taken on
 * the A operand's way into the bf16 split. */
int seg3d_linear_fwd_sum(const float* x, const float* x_add, int64_t m, const void* w_packed, const float* bias,
                         int32_t cin, int32_t cout, float* y, void* stream);
/* y = res + LayerNorm(x W^T + b) in ONE launch (inference path): the encoder layer's out-projection + norm1 + residual and
 * fc2 + norm2 + residual (point_transformer_layer.py:289-298; the reference runs Linear, LayerNorm and the add as three
 * kernels).  The normalised row must fit one workgroup: cout in {16, 32, 48, 64, 96, 128, 192}, SEG3D_EINVAL otherwise
 * (the caller then runs seg3d_linear_fwd + seg3d_layernorm_fwd).  res nullable.  Statistics as seg3d_layernorm_fwd takes
 * them: two-pass mean / biased variance in float32, rstd = rsqrt(var + eps). */
int seg3d_linear_layernorm_fwd(const float* x, int64_t m, const void* w_packed, const float* bias, const float* res,
                               const float* gamma, const float* beta, float eps, int32_t cin, int32_t cout, float* y,
                               void* stream);
/* y = (x W^T) * factor, elementwise ([m, cout] factor, no bias): the input gradient of the MLP's fc2 times the GELU
 * derivative saved by the training forward (point_transformer_layer.py:260-276; torch runs a gelu_backward pass there). */
int seg3d_linear_fwd_mul(const float* x, int64_t m, const void* w_packed, const float* factor, int32_t cin, int32_t cout,
                         float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * a13, a15, a16, a18  get_window_coors / batching_single_shift / get_flat2win_inds /
 *      make_continuous_inds -- seg3d/utils/swformer_utils.py:8-31,108-171,
 *      seg3d/models/layers/point_transformer_layer.py:71-87,141-149,209-220.
 * One call per (stage, shift).  Window id = b*Wx*Wy*Wz + wx*(Wy*Wz) + wy*Wz + wz with
 * w* = (c* + shift*) / win* (floor), in-window coordinate = (c* + shift*) % win*.
 *   win_xyz, nwin_xyz, shift_xyz : host int32[3]  (nwin = ceil(S/win)+1, swformer_utils.py:119-121)
 *   level_lo/level_hi/level_cap  : host int32[n_levels]  batching_range and max_tokens per level
 * Outputs (each [m] unless noted; any of win_id..slot may be NULL):
 *   win_id   batch_win_inds            in_win [m,3]  coors_in_win as (z,y,x)
 *   rank     in-window rank (ascending voxel row)    level  batching level, -1 if no range matches
 *   slot     flat2window index = compact_window_in_level * max_tokens + rank; -1 if rank >= max_tokens
 *   tok      voxel rows grouped by window (ascending window id, ascending row inside)
 *   win_start/win_count [<= min(m, canvas)]  CSR of the non-empty windows into tok
 *   win_tile0 [<= min(m, canvas)]  first 32-token tile of each window in the 32-padded token space
 *   tile_item [<= m/32 + windows][4]  {window, 32-token tile, win_start, win_count}: work items of the attention kernels, each
 *                                     a whole descriptor (one 16-byte read; no dependent win_start / win_count reads in front
 *                                     of a workgroup's first gather)
 *   qg_item   [<= m/16 + windows][4]  {window, 128-token chunk = 4 tiles, win_start, win_count}: items of the wide-head kernels
 *   counts   device int32[4]: {non-empty windows, voxels with slot == -1, 32-token tiles, 128-query chunks}
 */
size_t seg3d_window_partition_workspace_bytes(int64_t m, int32_t batch_size, const int32_t* nwin_xyz);
int seg3d_window_partition(const int32_t* coords, int64_t m, int32_t batch_size,
                           const int32_t* win_xyz, const int32_t* nwin_xyz, const int32_t* shift_xyz,
                           int32_t n_levels, const int32_t* level_lo, const int32_t* level_hi,
                           const int32_t* level_cap, int32_t* win_id, int32_t* in_win, int32_t* rank,
                           int32_t* level, int32_t* slot, int32_t* tok, int32_t* win_start,
                           int32_t* win_count, int32_t* win_tile0, int32_t* tile_item,
                           int32_t* qg_item, int32_t* counts, void* workspace,
                           size_t workspace_bytes, void* stream);

/* a17  SparseWindowPartitionLayer.get_pos_embed -- point_transformer_layer.py:151-203
 * pos[i] = cat_{d in x,y,z} interleave(sin(v_d / f[0::2]), cos(v_d / f[1::2])), v_d = in_win_d - win_d/2,
 * f = inv_freq [c/3] (device; the caller evaluates 1000**(2*(j//2)/(c/3)) exactly as torch does). */
int seg3d_pos_embed(const int32_t* in_win, int64_t m, const int32_t* win_xyz /*host[3]*/,
                    const float* inv_freq, int32_t c, float* pos, void* stream);

/* ------------------------------------------------------------------------------------------
 * a19-a21  flat2window -> CosineMultiheadAttention core -> window2flat --
 *      swformer_utils.py:34-85, point_transformer_layer.py:233-258,
 *      seg3d/models/layers/cosine_msa.py:115-177 (_scaled_cosine_attention).
 * Variable-length (CSR) form: no padded [W,T,C] tensors are materialised.  For every window w,
 * head h and query token i:  out_i = softmax_j( <q_i/|q_i|, k_j/|k_j|> / max(tau, tau_min) ) . v_j
 * over the tokens j of the same window.  q, k, v are the in-projection outputs in flat voxel
 * order with row strides ldq/ldk/ldv (floats); head h occupies columns [h*dh, (h+1)*dh).
 *   out [m, heads*dh] flat voxel order (input of the out-projection); lse [m, heads] or NULL
 *   (log-sum-exp per query row, kept for the backward).  m = number of voxel rows; dh in {6,12,24,48}
 *   (8 heads on 48/96/192/384 channels, pointtransformer.py:141-157).
 * Forward runs on the matrix cores in split-bf16 arithmetic (q, k, v, P as bf16 hi + lo, three
 * v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate and softmax) in ONE launch per layer of PERSISTENT workgroups:
 * each walks its share of the work items -- (32-query tile, 4 heads) for dh 6 / 12 (tile_item), (128-query chunk, head)
 * for dh 24 / 48 (qg_item: the second field counts 128-token chunks, four 32-token tiles) -- gathers, normalises and
 * splits the item's query rows and the window's k / v rows itself, 32 keys at a time, through LDS, and requests the next
 * item's rows while the current one is multiplied; n_tiles / n_qgroups are counts[2..3] of seg3d_window_partition.
 * dropout_p / dropout_seed: attention-probability dropout of training mode (cosine_msa.py:172-174); 0 = none.
 * The mask is a function of (seed, window, head, query, key) only -- the backward is given the same two values
 * and regenerates it.  Drop probability is rounded to a multiple of 1/256 and compensated exactly.
 * Backward takes the forward's out and lse, returns gradients w.r.t. the raw q, k, v (through the
 * normalisation) and stores the tau gradient in dtau[0] (per-wave partials in the workspace, summed in a
 * fixed order: the whole backward is free of atomics and identical from run to run).
 * Head geometries: dh 6 / 12 with a head count that is a multiple of 4 (dh 6: <= 8 heads), dh 24 / 48 with up to 16 heads
 * (the reference builds 8 heads of 6 / 12 / 24 / 48 channels, pointtransformer.py:143-155); seg3d_window_attn_supported
 * returns 1 for those, everything else is refused with SEG3D_EINVAL.  win_tile0 is accepted and ignored (it served
 * kernels removed in ABI 30); the forward needs no workspace, seg3d_window_attn_workspace_bytes sizes the backward's.
 */
int seg3d_window_attn_supported(int32_t heads, int32_t dh);
size_t seg3d_window_attn_workspace_bytes(int64_t m, int32_t n_tiles, int32_t heads, int32_t dh);
int seg3d_window_attn_fwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk,
                          int32_t ldv, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* win_tile0,
                          const int32_t* tile_item, int32_t n_tiles, const int32_t* qg_item,
                          int32_t n_qgroups, int64_t m, int32_t n_windows, int32_t heads, int32_t dh,
                          const float* tau, float tau_min, float dropout_p, uint64_t dropout_seed, float* out,
                          float* lse, void* workspace, size_t workspace_bytes, void* stream);
int seg3d_window_attn_bwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk,
                          int32_t ldv, const float* out, const float* dout, const float* lse,
                          const int32_t* tok, const int32_t* win_start, const int32_t* win_count,
                          const int32_t* win_tile0, const int32_t* tile_item, int32_t n_tiles,
                          const int32_t* qg_item, int32_t n_qgroups, int64_t m, int32_t n_windows,
                          int32_t heads, int32_t dh, const float* tau,
                          float tau_min, float dropout_p, uint64_t dropout_seed, float* dq, float* dk,
                          float* dv, int32_t lddq, int32_t lddk,
                          int32_t lddv, float* dtau, void* workspace, size_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * a12, a22  row-wise normalisation layers on [m, c] features (c multiple of 4).
 * LayerNorm of the post-norm encoder layer (point_transformer_layer.py:289-298), fused with the residual:
 *     y = res + rowscale_row * ((x - mean_row) * rstd_row * gamma + beta)
 * res may be NULL; rowscale [m] (the per-row DropPath factor, seg3d/models/layers/drop.py:6-19) may be NULL = 1;
 * mean/rstd [m] are kept for the backward pass.
 * backward: dx, and dgamma/dbeta [c] (per-block partial sums in the workspace, summed in a fixed order: deterministic);
 * the residual's gradient is dy itself.
 * BatchNorm1d (+ residual) (+ ReLU) of the conv blocks / point MLPs (spconv_utils.py:13-32,
 * pointtransformer.py:47-66, segformer.py:21-76):
 *     seg3d_colstats   sums[0..c) = sum_r (x - x[0]), sums[c..2c) = sum_r (x - x[0])^2   (shifted, cancellation-free)
 *                      per-block partial sums in the workspace (seg3d_batchnorm_workspace_bytes), summed in a fixed
 *                      order: no atomics, results are identical from run to run
 *     seg3d_batchnorm_stats  colstats + one finalize pass: stats [6][c] = {scratch, scratch, mean, rstd (biased
 *                      variance), scale = gamma*rstd, shift = beta - mean*scale}; running_mean / running_var (may be
 *                      NULL) updated in place with `momentum` (unbiased variance), as torch.nn.BatchNorm1d does
 *     seg3d_affine_act y = act(x * scale + shift (+ res)), scale/shift [c] as produced above (or folded from the
 *                      running statistics by the caller in eval mode)
 *     seg3d_batchnorm_bwd  g = dy masked by (y > 0) when relu; dres = g (may be NULL);
 *                      dx = gamma*rstd*(g - mean_r(g) - xhat*mean_r(g*xhat)); sums = {dbeta, dgamma}
 */
int seg3d_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                        const float* rowscale, float eps, int64_t m, int32_t c, float* y, float* mean,
                        float* rstd, void* stream);
size_t seg3d_layernorm_bwd_workspace_bytes(int64_t m, int32_t c);
int seg3d_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                        const float* gamma, const float* rowscale, int64_t m, int32_t c, float* dx,
                        float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* seg3d_layernorm_bwd without its final sum: dx now, the per-block partial sums of dgamma / dbeta left in the workspace as
 * part[*nblocks][2][c] -- a seg3d_reduce_partials job with n = 2 c, nw = c, dw = dgamma, db = dbeta. */
int seg3d_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* rowscale, int64_t m, int32_t c, float* dx,
                                 void* workspace, size_t workspace_bytes, int32_t* nblocks, void* stream);
size_t seg3d_batchnorm_workspace_bytes(int64_t m, int32_t c);
int seg3d_colstats(const float* x, int64_t m, int32_t c, float* sums, void* workspace, size_t workspace_bytes,
                   void* stream);
int seg3d_batchnorm_stats(const float* x, int64_t m, int32_t c, float eps, const float* gamma, const float* beta,
                          float momentum, float* running_mean, float* running_var, float* stats, void* workspace,
                          size_t workspace_bytes, void* stream);
/* GELU, erf form (torch.nn.GELU default, point_transformer_layer.py:265): g = h Phi(h); gp (nullable) = d g / d h =
 * Phi(h) + h phi(h), written in the same pass by the training forward.  n = number of elements, a multiple of 4. */
int seg3d_gelu_fwd(const float* h, int64_t n, float* g, float* gp, void* stream);
int seg3d_affine_act(const float* x, const float* res, const float* scale, const float* shift, int32_t relu,
                     int64_t m, int32_t c, float* y, void* stream);
int seg3d_batchnorm_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                        const float* gamma, int32_t relu, int64_t m, int32_t c, float* dx, float* dres,
                        float* sums, void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.SyncBatchNorm (tools/train.py:246-247, --sync_bn): the two halves of seg3d_batchnorm_bwd as separate
 * entries.  _reduce writes the rank-local sums = {sum g, sum g*xhat} (= dbeta, dgamma of this rank, which is what
 * torch's SyncBatchNorm hands DDP); the caller all-reduces a copy over the ranks and passes it to _apply together
 * with inv_count = device float[1] holding 1 / (rows of all ranks); mean / rstd are the synchronised statistics. */
int seg3d_batchnorm_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean,
                               const float* rstd, int32_t relu, int64_t m, int32_t c, float* sums,
                               void* workspace, size_t workspace_bytes, void* stream);
int seg3d_batchnorm_bwd_apply(const float* dy, const float* y, const float* x, const float* mean,
                              const float* rstd, const float* gamma, const float* sums, const float* inv_count,
                              int32_t relu, int64_t m, int32_t c, float* dx, float* dres, void* stream);

/* ------------------------------------------------------------------------------------------
 * a7, a25, a26  torch_scatter.scatter(src, index, dim=0, reduce='mean'|'max') at
 *      seg3d/models/voxel_encoders/vfe.py:24-25, seg3d/models/layers/se_layer.py:24-28, and
 *      voxel_pooling_ext.voxel_pooling_{forward,backward}_* -- seg3d/ops/voxel_pooling/src/
 *      voxel_pooling.cpp:5-43, voxel_pooling_cuda.cu:10-79.
 * Segmented reduce over a CSR (order, offsets) built by seg3d_group_index: deterministic, no
 * float atomics.  Empty segments give 0.  argmax [n_seg, c] (MAX only, may be NULL) records the
 * source row (first maximum in ascending row order), -1 for empty segments.
 */
int seg3d_segment_reduce_fwd(const float* x, int32_t c, const int32_t* order, const int32_t* offsets,
                             int64_t n_seg, int32_t mode, float* out, int32_t* argmax, void* stream);
/* dx [n, c] must be zero-filled by the caller for MAX; fully written for SUM/MEAN rows that belong
 * to a segment (seg_of_row = group ids, -1 rows get 0). */
int seg3d_segment_reduce_bwd(const float* dout, int32_t c, const int32_t* seg_of_row, int64_t n,
                             const int32_t* offsets, const int32_t* argmax, int64_t n_seg,
                             int32_t mode, float* dx, void* stream);

/* SURVEY 8(f) rank 1  DeepFusionBlock's cross attention over the kNN rows -- seg3d/models/layers/deep_fusion.py:26-45:
 *     out_i = sum_j dropout(nan_to_num(softmax_j(<q_i, k_{idx[i][j]}> * scale, -inf where invalid[idx[i][j]]))) . v_{idx[i][j]}
 * q [n, d], k / v [n_src, d], idx int32 [n, n_neighbors] (rows of k / v), d = 32, n_neighbors <= 16, scale = 1 / sqrt(d).
 * invalid uint8 [n_src] (1 = the source row has no image feature: torch.sum(image_features, 1) == 0) or NULL;
 * keep float [n, n_neighbors] = the dropout factors F.dropout would multiply with (0 or 1 / (1 - p)) or NULL;
 * prob [n, n_neighbors] receives the softmax (before dropout) for the backward, may be NULL.  Nothing of size
 * [n, n_neighbors, d] is materialised.  Backward: dq, and dk / dv by a fixed-order sum over the inverse neighbour
 * lists -- pair_order / pair_offsets = seg3d_group_index of the flattened idx table over the n_src rows -- so it is
 * free of float atomics and bit-reproducible; scratch holds 2 * n * n_neighbors floats. */
int seg3d_knn_attention_fwd(const float* q, const float* k, const float* v, const int32_t* idx,
                            const uint8_t* invalid, const float* keep, int64_t n, int64_t n_src,
                            int32_t n_neighbors, int32_t d, float scale, float* out, float* prob, void* stream);
int seg3d_knn_attention_bwd(const float* q, const float* k, const float* v, const int32_t* idx, const float* keep,
                            const float* prob, const float* dout, const int32_t* pair_order,
                            const int32_t* pair_offsets, int64_t n, int64_t n_src, int32_t n_neighbors, int32_t d,
                            float scale, float* dq, float* dk, float* dv, float* scratch, void* stream);

/* SURVEY 8(f) rank 3  OCR's SpatialGatherModule -- seg3d/models/layers/ocr.py:10-36: per sample b (a span of rows:
 * offsets [batch] = cumulative row counts, the stride-8 level is sorted by batch index) and class k,
 *     context[b, k, :] = sum_{r in b} softmax_r(scale * probs[r, k]) * feats[r, :]
 * instead of a Python loop over the samples with boolean masks.  feats [m, c], probs [m, classes <= 32];
 * chunks [n_chunks][2] = (sample, first row) of every 128-row chunk in sample order, chunk_offsets [batch] = cumulative
 * chunk counts.  weights [m, classes] (kept for the backward), partials [n_chunks, classes, c] and stats
 * [batch, classes, 2] are caller-provided scratch; sums run in a fixed order (no atomics).  Backward returns d feats
 * [m, c] and d probs [m, classes]; scratch = m * classes + n_chunks * classes floats. */
int seg3d_class_context_fwd(const float* feats, const float* probs, const int32_t* offsets, const int32_t* chunks,
                            const int32_t* chunk_offsets, int32_t n_chunks, int32_t batch, int64_t m,
                            int32_t classes, int32_t c, float scale, float* weights, float* partials, float* stats,
                            float* context, void* stream);
int seg3d_class_context_bwd(const float* feats, const float* weights, const float* dcontext,
                            const int32_t* offsets, const int32_t* chunks, const int32_t* chunk_offsets,
                            int32_t n_chunks, int32_t batch, int64_t m, int32_t classes, int32_t c, float scale,
                            float* dfeats, float* dprobs, float* scratch, void* stream);

/* a24  VoxelToPoint.__call__ -- seg3d/ops/voxel_to_point/voxel_to_point.py:4-17
 * out[i] = feats[ids[i]] (zeros where ids[i] == -1).  Its backward is
 * seg3d_segment_reduce_fwd(dout, SUM) over the CSR of ids. */
int seg3d_gather_rows(const float* feats, const int32_t* ids, int64_t n, int32_t c, float* out,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f) rank 1  knn_query_ext.knn_query_cuda -- seg3d/ops/knn_query/src/knn_query_cuda.cu:67-133,
 *      wrapper seg3d/ops/knn_query/knn_query.py:7-24 (callers: DeepFusionBlock deep_fusion.py:31, tools/train.py:103).
 * For every query row of new_xyz [m,3] the k nearest rows of xyz [n,3] inside the same batch segment
 * (offset / new_offset: cumulative int32 counts per sample), ascending squared distance; ties by ascending
 * candidate index; slots beyond the segment size hold (1e10, segment start) as in the reference.  k <= 64.
 */
int seg3d_knn_query(const float* xyz, int64_t n, const float* new_xyz, int64_t m, const int32_t* offset,
                    const int32_t* new_offset, int32_t batch_size, int32_t k, int32_t* idx, float* dist2,
                    void* stream);

/* Grid-accelerated exact kNN (same results and tie rule as seg3d_knn_query) for large clouds.  Nothing here reads a
 * device value on the host: a whole query is a fixed sequence of launches.
 *   seg3d_knn_level_build  one grid level over xyz [n,3]: key = (batch << 48 | cx << 32 | cy << 16 | cz), c* =
 *                          floor(p / cell) + 32768 (clamped); device radix sort of (key, row); sorted_xyz [n,3] = points
 *                          in cell order, src_index [n] = their original rows; for the first point of every cell
 *                          (sorted position s): cell_end[s] = end of the cell's run, and key -> s goes into the
 *                          open-addressing table (capacity: power of two >= 2 n; table_keys capacity x 8 B,
 *                          table_vals capacity x 4 B).  Workspace: seg3d_knn_level_workspace_bytes(n) (0 = invalid n or
 *                          no device visible -- the sort's scratch size comes from rocPRIM).
 *   seg3d_knn_query_order  order [m] = query rows sorted by their cell key on a grid of the given cell size (neighbouring
 *                          lanes then walk the same cells); same workspace query with n = m.
 *   seg3d_knn_grid_query   per query: cube shells around its cell until the k-th distance is below the shell bound; up to
 *                          four grids, fine to coarse (lidar density falls as 1/r^2): a query still open after a level's
 *                          max_ring shells restarts on the next level, and scans its whole segment after the last one.
 *                          When every level's cell is exactly 8x the previous one, a coarse cell holding > 128 points is
 *                          searched through its sub-cells, pruned by box distance against the current k-th distance.
 *                          batch_size <= 255.  `levels` is a HOST array. */
size_t seg3d_knn_level_workspace_bytes(int64_t n);
int seg3d_knn_level_build(const float* xyz, int64_t n, const int32_t* offset, int32_t batch_size, float cell,
                          float* sorted_xyz, int32_t* src_index, int32_t* cell_end, void* table_keys, int32_t* table_vals,
                          int64_t capacity, void* workspace, size_t workspace_bytes, void* stream);
int seg3d_knn_query_order(const float* new_xyz, int64_t m, const int32_t* new_offset, int32_t batch_size, float cell,
                          int32_t* order, void* workspace, size_t workspace_bytes, void* stream);
/* a10/a11 scheduling aid: order [m] = rows of coords [m,4] (b,z,y,x) grouped by the parity of (z, y, x), original order
 * kept inside a group (stable 3-bit device radix sort).  Under SparseConv3d(k=3, s=2, p=1) the kernel offsets that can
 * reach a fine site are fixed by its parity, so 128-row tiles of same-parity rows of the inverse table visit 1-8 of
 * the 27 offsets (row_order of seg3d_spconv_fwd).  Workspace: seg3d_knn_level_workspace_bytes(m). */
int seg3d_parity_order(const int32_t* coords, int64_t m, int32_t* order, void* workspace, size_t workspace_bytes,
                       void* stream);
typedef struct {
  const float* sorted_xyz;     /* [n,3] points in this level's cell order */
  const int32_t* src_index;    /* [n] original row of each sorted point */
  const int32_t* cell_end;     /* [n] valid at the first sorted position of every cell */
  const void* table_keys;      /* capacity x 8 B */
  const int32_t* table_vals;   /* capacity x 4 B: first sorted position of the cell */
  int64_t capacity;
  float cell;                  /* metres */
  int32_t max_ring;            /* shells walked on this level (<= 16) */
} seg3d_knn_level;
int seg3d_knn_grid_query(const seg3d_knn_level* levels, int32_t n_levels, const float* new_xyz,
                         const int32_t* query_order /* queries in level-0 cell order, or NULL */, int64_t m,
                         const int32_t* offset, const int32_t* new_offset, int32_t batch_size, int32_t k,
                         int32_t* idx, float* dist2, void* stream);

/*
 * SURVEY 8(f) rank 2  WaymoDataset.prepare_voxel_labels (seg3d/datasets/waymo_dataset.py:213-246): label of a voxel =
 * most frequent label (uint8, 0..255, the ignore label counted like any other) among its points, ties to the smallest
 * label, ignore_index for voxels without a point.  order / offsets: the point->voxel CSR of seg3d_group_index
 * (points with id -1 -- dropped or history-sweep points -- are in no segment).
 */
int seg3d_voxel_majority_labels(const uint8_t* point_labels, const int32_t* order, const int32_t* offsets,
                                int64_t n_voxels, int32_t ignore_index, uint8_t* voxel_labels, void* stream);

/*
 * SURVEY 8(f) rank 4  'ce' / 'ohem_ce' terms of build_criterion (seg3d/models/builder.py:26-40) on logits [n, c] with
 * int64 labels; rows whose label is ignore_index (or outside [0, c)) contribute nothing.
 *   keep_thresh <= 0 : nn.CrossEntropyLoss(ignore_index), mean over the counted rows
 *   keep_thresh  > 0 : OHEMCrossEntropyLoss(keep_thresh) (seg3d/models/losses/ohem_cross_entropy_loss.py:23-38): only
 *                      rows with softmax(logits)[label] < keep_thresh are counted
 * forward: lse [n] kept for backward (log-sum-exp on counted rows, +inf on the others), stats = {mean loss, count}
 * (mean = 0 when nothing is counted; torch returns NaN there); backward: dlogits = (softmax - onehot) * grad_out / count
 * on counted rows, 0 elsewhere.  Deterministic (per-block partials, fixed-order finalize).
 */
size_t seg3d_cross_entropy_workspace_bytes(int64_t n);
int seg3d_cross_entropy_fwd(const float* logits, const int64_t* labels, int64_t n, int32_t c, int64_t ignore_index,
                            float keep_thresh, float* lse, float* stats, void* workspace, size_t workspace_bytes,
                            void* stream);
int seg3d_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* stats,
                            const float* grad_out, int64_t n, int32_t c, float* dlogits, void* stream);

/*
 * SURVEY 8(f) rank 4  Lovasz-softmax term: LovaszLoss(ignore_index) as build_criterion makes it (multi_class, per_image
 * False; seg3d/models/losses/lovasz_loss.py:13-26 lovasz_grad, :118-158 lovasz_softmax_flat, :268-290 forward) on logits
 * [n, c <= 64] with int64 labels, n * c < 2^31.  All classes go through ONE device radix sort of (class, error) keys
 * instead of c separate sorts; the jaccard differences use the reference's float32 arithmetic.
 *   classes_mode 0 = 'present' (classes without a label in the batch are skipped), 1 = 'all'; include (nullable int32
 *   [c], non-zero = averaged) is the explicit class list of the reference and overrides the presence test; class_weight
 *   nullable float [c].  Rows with label == ignore_index take no part.
 * forward : coef [n, c] (d loss_class / d prob, kept for backward), stats [2 + c] = {loss, classes averaged,
 *           d loss / d loss_class per class}
 * backward: dlogits [n, c] = softmax backward of coef * stats[2 + class] * grad_out[0]
 * seg3d_lovasz_workspace_bytes asks rocPRIM for the sort's scratch size on the current device: 0 = invalid arguments or
 * no device visible.
 */
size_t seg3d_lovasz_workspace_bytes(int64_t n, int32_t c);
int seg3d_lovasz_softmax_fwd(const float* logits, const int64_t* labels, int64_t n, int32_t c, int64_t ignore_index,
                             int32_t classes_mode, const int32_t* include, const float* class_weight, float* coef,
                             float* stats, void* workspace, size_t workspace_bytes, void* stream);
int seg3d_lovasz_softmax_bwd(const float* logits, const float* coef, const float* stats, const float* grad_out,
                             int64_t n, int32_t c, float* dlogits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEG3D_HIP_H */

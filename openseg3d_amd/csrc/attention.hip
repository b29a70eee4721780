// a19-a21: C-ABI entry points of the ragged sparse-window cosine attention (include/seg3d_hip.h) and the choice of
// kernels behind them.  Reference: flat2window -> CosineMultiheadAttention -> window2flat (swformer_utils.py:34-85,
// point_transformer_layer.py:233-258, cosine_msa.py:115-177).
//   forward   attention_fused.hip      persistent fused kernel, every head width (dh 6 / 12 / 24 / 48), dropout-capable
//             attention_small.hip      exact-fp32 vector-ALU kernel, dh 6 with dropout (round 4: 101 / 96 -> 86 / 73 us per layer)
//   backward  attention_fused_bwd.hip  two flash-style passes, dh 12 / 24 / 48
//             attention_small.hip      exact-fp32 vector-ALU passes, dh 6 (windows of ~15 voxels: 360 vs 475 us per layer)
// Head geometries these kernels do not take (narrow heads whose count is not a multiple of 4, head widths other than the
// four the reference builds -- pointtransformer.py:143-155: 8 heads of 6 / 12 / 24 / 48 channels) are refused with
// SEG3D_EINVAL; seg3d_window_attn_supported says so up front.  The round-1 / round-2 fallbacks (prepare pass + core,
// three-launch backward with prepared operands through HBM, vector-ALU forward) were removed in round 3: nothing on the
// reference's path reached them and their workspace demand (gigabytes on a 2 M-point scene) leaked into every call.
#include <cstdlib>

#include "attn_common.hpp"
#include "attn_dropout.hpp"

bool attn_fused_supported(int heads, int dh);  // attention_fused.hip
int attn_fused_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles,
                          const int32_t* chunk_item, int n_chunks, int heads, int dh, const float* tau, float tau_min,
                          float* out, float* lse, float dropout_p, uint64_t seed, hipStream_t st);
size_t attn_fused_bwd_workspace_bytes(int64_t m, int n_tiles, int n_chunks, int heads, int dh);  // attention_fused_bwd.hip
bool attn_fused_bwd_supported(int heads, int dh);
int attn_fused_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, const int32_t* chunk_item,
                          int n_chunks, int64_t m, int heads, int dh, const float* tau, float tau_min, float* dq, float* dk,
                          float* dv, int lddq, int lddk, int lddv, float* dtau, void* workspace, const DropoutParams& drop,
                          hipStream_t st);
bool attn_small_supported(int heads, int dh);  // attention_small.hip
int attn_small_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles, int heads,
                          int dh, const float* tau, float tau_min, float* out, float* lse, const DropoutParams& drop,
                          hipStream_t st);
int attn_small_bwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const float* out,
                          const float* dout, const float* lse, const int32_t* tok, const int32_t* win_start,
                          const int32_t* win_count, const int32_t* tile_item, int n_tiles, int heads, int dh,
                          const float* tau, float tau_min, float* dq, float* dk, float* dv, int lddq, int lddk, int lddv,
                          float* dtau, void* workspace, const DropoutParams& drop, hipStream_t st);

namespace {

// backward kernels for a head geometry: 0 fused (two flash-style passes), 1 vector-ALU (dh 6), -1 none
int bwd_path(int heads, int dh) {
    if (!attn_fused_supported(heads, dh)) return -1;
    if (attn_fused_bwd_supported(heads, dh)) return 0;
    if (attn_small_supported(heads, dh)) return 1;
    return -1;
}

bool rows_misaligned(const void* a, const void* b, const void* c, int lda, int ldb, int ldc) {
    return ((lda | ldb | ldc) & 3) ||
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15);
}

}  // namespace

extern "C" {

int seg3d_window_attn_supported(int32_t heads, int32_t dh) {
    return heads > 0 && heads <= 16 && bwd_path(heads, dh) >= 0 ? 1 : 0;
}

// Bytes for the kernels seg3d_window_attn_fwd / _bwd take with these arguments: nothing in the forward, one tau-gradient
// partial per wave (fused) or per tile (vector-ALU) in the backward, plus <dO, O> per (token, head) between its two passes.
size_t seg3d_window_attn_workspace_bytes(int64_t m, int32_t n_tiles, int32_t heads, int32_t dh) {
    if (m < 0 || heads <= 0 || n_tiles < 0) return 0;
    size_t bwd = 0;
    switch (bwd_path(heads, dh)) {
        case 0: bwd = attn_fused_bwd_workspace_bytes(m, n_tiles, n_tiles, heads, dh); break;  // chunks <= tiles
        case 1: bwd = (size_t)n_tiles * sizeof(float); break;
        default: return 0;
    }
    return bwd + 256;
}

int seg3d_window_attn_fwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                          const int32_t* tok, const int32_t* win_start, const int32_t* win_count,
                          const int32_t* win_tile0, const int32_t* tile_item, int32_t n_tiles, const int32_t* qg_item,
                          int32_t n_qgroups, int64_t m, int32_t n_windows, int32_t heads, int32_t dh, const float* tau,
                          float tau_min, float dropout_p, uint64_t dropout_seed, float* out, float* lse, void* workspace,
                          size_t workspace_bytes, void* stream) {
    (void)win_tile0;
    (void)workspace;
    (void)workspace_bytes;
    if (m == 0 || n_windows == 0 || n_tiles == 0 || n_qgroups == 0) return SEG3D_OK;
    if (!q || !k || !v || !tok || !win_start || !win_count || !tile_item || !qg_item || m < 0 || n_windows < 0 ||
        n_tiles < 0 || n_qgroups < 0 || heads <= 0 || heads > 16 || !tau || !out || !(dropout_p >= 0.f && dropout_p < 1.f))
        return SEG3D_EINVAL;
    if (!attn_fused_supported(heads, dh)) return SEG3D_EINVAL;
    if (rows_misaligned(q, k, v, ldq, ldk, ldv)) return SEG3D_EINVAL;  // rows are gathered in 16-B pieces
    // dh 6 (stage 1: windows of ~15 voxels) WITH dropout: exact fp32 on the vector ALUs, like its backward -- one hash per key
    // pair and thread instead of the MFMA kernel's per-fragment mask: 101 -> 86 / 96 -> 73 us per layer (shift 0 / 1).  Without
    // dropout the two forms tie (84 / 67 against 72 / 66 us) and the fused kernel stays.  SEG3D_ATTN_SMALL_FWD=0 | 1 | 2 (A/B):
    // never / with dropout (default) / always.
    static const int small_fwd = getenv("SEG3D_ATTN_SMALL_FWD") ? atoi(getenv("SEG3D_ATTN_SMALL_FWD")) : 1;
    if ((small_fwd == 2 || (small_fwd == 1 && dropout_p > 0.f)) && attn_small_supported(heads, dh) && lse)
        return attn_small_fwd_launch(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, tile_item, n_tiles, heads, dh, tau, tau_min,
                                     out, lse, make_dropout(dropout_p, dropout_seed), as_stream(stream));
    return attn_fused_fwd_launch(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, tile_item, n_tiles, qg_item, n_qgroups,
                                 heads, dh, tau, tau_min, out, lse, dropout_p, dropout_seed, as_stream(stream));
}

int seg3d_window_attn_bwd(const float* q, const float* k, const float* v, int32_t ldq, int32_t ldk, int32_t ldv,
                          const float* out, const float* dout, const float* lse, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* win_tile0,
                          const int32_t* tile_item, int32_t n_tiles, const int32_t* qg_item, int32_t n_qgroups, int64_t m,
                          int32_t n_windows, int32_t heads, int32_t dh, const float* tau, float tau_min, float dropout_p,
                          uint64_t dropout_seed, float* dq, float* dk, float* dv, int32_t lddq, int32_t lddk, int32_t lddv,
                          float* dtau, void* workspace, size_t workspace_bytes, void* stream) {
    (void)win_tile0;
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return SEG3D_EINVAL;
    const DropoutParams drop = make_dropout(dropout_p, dropout_seed);  // same mask as the forward, given the same two values
    if (m == 0 || n_windows == 0 || n_tiles == 0 || n_qgroups == 0) {
        if (dtau) SEG3D_CHECK_HIP(hipMemsetAsync(dtau, 0, sizeof(float), as_stream(stream)));
        return SEG3D_OK;
    }
    if (!q || !k || !v || !out || !dout || !lse || !tok || !win_start || !win_count || !tile_item || !qg_item || m < 0 ||
        n_windows < 0 || n_tiles < 0 || n_qgroups < 0 || heads <= 0 || heads > 16 || !tau || !dq || !dk || !dv || !dtau ||
        !workspace)
        return SEG3D_EINVAL;
    if (rows_misaligned(q, k, v, ldq, ldk, ldv) || rows_misaligned(dq, dk, dv, lddq, lddk, lddv))
        return SEG3D_EINVAL;  // rows are gathered / stored in 16-B pieces
    const int path = bwd_path(heads, dh);
    if (path < 0) return SEG3D_EINVAL;
    if (path == 0) {
        if (workspace_bytes < attn_fused_bwd_workspace_bytes(m, n_tiles, n_qgroups, heads, dh)) return SEG3D_EWORKSPACE;
        return attn_fused_bwd_launch(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, tile_item, n_tiles,
                                     qg_item, n_qgroups, m, heads, dh, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau,
                                     workspace, drop, as_stream(stream));
    }
    if (workspace_bytes < (size_t)n_tiles * sizeof(float)) return SEG3D_EWORKSPACE;
    return attn_small_bwd_launch(q, k, v, ldq, ldk, ldv, out, dout, lse, tok, win_start, win_count, tile_item, n_tiles, heads,
                                 dh, tau, tau_min, dq, dk, dv, lddq, lddk, lddv, dtau, workspace, drop, as_stream(stream));
}

}  // extern "C"

// a19-a21 forward, fused: ragged sparse-window cosine attention in ONE launch per layer.
// Reference: flat2window -> CosineMultiheadAttention -> window2flat (swformer_utils.py:34-85,
// point_transformer_layer.py:233-258, cosine_msa.py:115-177).
//
// What the earlier forward did in two launches (attn_prepare_fwd: gather + L2-normalise + bf16 hi/lo split + transposed
// copies through HBM; attn_core_fwd_lds: the MFMA core) happens here inside the core's own staging step:
//   * a workgroup (4 waves) owns the queries of one work item and walks the window's 32-key tiles.  Per key tile all 256
//     threads gather the raw fp32 k / v rows (4 threads per key), normalise k, split everything to bf16 hi + lo and park
//     the tile in LDS ROW-major ([key][head][channel]) -- K is read back as MFMA A fragments with ds_read_b128, V^T
//     fragments come out of the same row-major image through ds_read_b64_tr_b16 (the hardware transpose read), so nothing
//     is ever written transposed and nothing makes a round trip through HBM;
//   * S^T = K.Q^T (keys on the accumulator rows): a lane's 8 scores belong to ONE query column, and the accumulator of
//     the score MFMAs is, register for register, the B operand of O^T += V^T.P (tokens are consumed in the permuted
//     order kappa(g, j) the two tr-reads deliver);
//   * cosine attention bounds every score by log2(e) / max(tau, tau_min): the softmax uses that bound as a FIXED maximum
//     (no running max, no rescaling of O, and the bound enters as the initial accumulator of the score MFMAs, so
//     p = exp2(acc) directly); the row sum comes out of the PV product for free through a column of ones stored behind
//     each head's V channels.  Rows whose scores could underflow under the fixed bound (tau < ~0.036) take the online
//     max / rescale form instead -- a wave-uniform branch on the device scalar tau, no host decision;
//   * narrow heads (dh 6 / 12: windows of 13-60 voxels) put all 8 heads of a 32-query tile in one workgroup (whole rows
//     are gathered once, two heads per wave); wide heads (dh 24 / 48) put four 32-query tiles of one head in a workgroup.
// Arithmetic: split-bf16 products (hi*hi + hi*lo + lo*hi), fp32 accumulate, as everywhere else on this path.
// Training: attention-probability dropout (cosine_msa.py:172-174) is applied to P inside the same loop from a counter-
// based hash of (seed, window, head, query, key); the backward regenerates the same mask (attn_dropout.hpp).
#include <type_traits>

#include "attn_fused.hpp"

#ifdef SEG3D_ATTN_STAMP
// Diagnostic build only (tools/probes/attn_stamps.py): per-wave s_memtime sums of the forward's phases.
__device__ unsigned long long* g_attn_stamp_buf = nullptr;
extern "C" int seg3d_debug_attn_stamps(void* buf) {
    unsigned long long* p = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : 2;
}
#define ASTAMP(i)                                                    \
    do {                                                             \
        __builtin_amdgcn_sched_barrier(0);                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        __builtin_amdgcn_sched_barrier(0);                           \
        st_acc[i] += t_ - st_last;                                   \
        st_last = t_;                                                \
    } while (0)
#define AKEEP(x) asm volatile("" ::"v"(x))
#else
#define ASTAMP(i) do {} while (0)
#define AKEEP(x) do {} while (0)
#endif

namespace {

using namespace attn;
using namespace attn_fused;

template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, (DROPOUT && Cfg<DH>::kWaves > 3 ? 3 : Cfg<DH>::kWaves)) void attn_fused_fwd(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ldq, int ldk, int ldv,
    const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
    const int2* __restrict__ items, int heads, const float* __restrict__ tau, float tau_min, float* __restrict__ out,
    float* __restrict__ lse, DropoutParams drop) {
    using C = Cfg<DH>;
    constexpr int HG = C::HG, QT = C::QT, UW = C::UW, DHS = C::DHS, KS = C::KS, VW = C::VW, NB = C::NB;
    constexpr int KRS = C::KRS, VRS = C::VRS, CT = C::CT;
    __shared__ __attribute__((aligned(16))) char lds[C::NBUF * C::kTile];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
#ifdef SEG3D_ATTN_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    const int2 item = items[blockIdx.x];
    const int n = win_count[item.x], start = win_start[item.x];
    AKEEP(n); AKEEP(start);
    ASTAMP(0);  // item record + window geometry (two dependent round trips)
    const int n_kt = (n + 31) >> 5;
    const int h0 = blockIdx.y * HG;
    const float qscale = kLog2e / fmaxf(tau[0], tau_min);
    // every score is <= qscale; p = exp2(s - qscale) >= 2^(-2 qscale) must stay a normal float
    const bool fixed_max = qscale <= 40.0f;

    // ---------------------------------------------------------------- staging role of this thread
    const int st_which = tid >> 7;            // 0: K rows, 1: V rows
    const int st_key = (tid & 127) >> 2, st_part = tid & 3;
    const float* st_src = st_which == 0 ? k : v;
    const int st_ld = st_which == 0 ? ldk : ldv;
    const int st_col = C::kNarrow ? (h0 + C::HPT * st_part) * DH : h0 * DH + st_part * CT;  // first channel of this thread's share
    float st_reg[CT];
    // token row of this thread's key in tile t: loaded a tile ahead of the rows, so that the row gather is one memory
    // round trip, not two dependent ones
    auto load_tok = [&](int t) {
        int kk = t * 32 + st_key;
        kk = kk < n ? kk : n - 1;  // clamped rows are finite and masked by p = 0
        return tok[start + kk];
    };
    auto stage_load = [&](int32_t token_row) {
        const float* row = st_src + (int64_t)token_row * st_ld + st_col;
        // widest aligned pieces: the share starts at a multiple of CT floats past a 16-B aligned head-group base
        if constexpr ((CT * 4) % 16 == 0 && (C::kNarrow ? (C::HPT * DH * 4) % 16 == 0 : (DH * 4) % 16 == 0)) {
#pragma unroll
            for (int i = 0; i < CT / 4; ++i) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * i);
                st_reg[4 * i] = x[0]; st_reg[4 * i + 1] = x[1]; st_reg[4 * i + 2] = x[2]; st_reg[4 * i + 3] = x[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CT / 2; ++i) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(row + 2 * i);
                st_reg[2 * i] = x[0]; st_reg[2 * i + 1] = x[1];
            }
        }
    };
    auto stage_store = [&](int buf) {
        char* base = lds + buf * C::kTile;
        if (st_which == 0) {  // K: L2-normalise per head, split, row-major [key][head][DHS]
            char* dst = base + st_key * KRS;
            if constexpr (C::kNarrow) {
#pragma unroll
                for (int hh = 0; hh < C::HPT; ++hh) {
                    float ss = 0.f;
#pragma unroll
                    for (int d = 0; d < DH; ++d) ss = fmaf(st_reg[hh * DH + d], st_reg[hh * DH + d], ss);
                    const float r = inv_norm(ss);
                    uint32_t hi[DHS / 2], lo[DHS / 2];
#pragma unroll
                    for (int i = 0; i < DHS / 2; ++i) {
                        const float a = 2 * i < DH ? st_reg[hh * DH + 2 * i] * r : 0.f;
                        const float b = 2 * i + 1 < DH ? st_reg[hh * DH + 2 * i + 1] * r : 0.f;
                        split2(a, b, &hi[i], &lo[i]);
                    }
                    char* p = dst + ((C::HPT * st_part + hh) * DHS) * 2;
#pragma unroll
                    for (int i = 0; i < DHS / 8; ++i) {
                        *reinterpret_cast<u32x4*>(p + 16 * i) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                        *reinterpret_cast<u32x4*>(p + C::kPlane + 16 * i) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                    }
                }
            } else {
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < CT; ++d) ss = fmaf(st_reg[d], st_reg[d], ss);
                ss = quad_sum(ss);  // the 4 threads of a key hold a quarter of the head each
                const float r = inv_norm(ss);
                char* p = dst + (st_part * CT) * 2;
#pragma unroll
                for (int i = 0; i < CT / 2; ++i) {
                    uint32_t hi, lo;
                    split2(st_reg[2 * i] * r, st_reg[2 * i + 1] * r, &hi, &lo);
                    *reinterpret_cast<uint32_t*>(p + 4 * i) = hi;
                    *reinterpret_cast<uint32_t*>(p + C::kPlane + 4 * i) = lo;
                }
            }
        } else {  // V: split only; [key][head][VW] with a 1.0 behind each head's channels (row sum of P for free)
            char* dst = base + 32 * KRS + st_key * VRS;
            if constexpr (C::kNarrow) {
#pragma unroll
                for (int hh = 0; hh < C::HPT; ++hh) {
                    uint32_t hi[VW / 2], lo[VW / 2];
#pragma unroll
                    for (int i = 0; i < VW / 2; ++i) {
                        const float a = 2 * i < DH ? st_reg[hh * DH + 2 * i] : (2 * i == DH ? 1.0f : 0.f);
                        const float b = 2 * i + 1 < DH ? st_reg[hh * DH + 2 * i + 1] : (2 * i + 1 == DH ? 1.0f : 0.f);
                        split2(a, b, &hi[i], &lo[i]);
                    }
                    char* p = dst + ((C::HPT * st_part + hh) * VW) * 2;
#pragma unroll
                    for (int i = 0; i < VW / 8; ++i) {
                        *reinterpret_cast<u32x4*>(p + 16 * i) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                        *reinterpret_cast<u32x4*>(p + C::kPlane + 16 * i) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                    }
                }
            } else {
                char* p = dst + (st_part * CT) * 2;
#pragma unroll
                for (int i = 0; i < CT / 2; ++i) {
                    uint32_t hi, lo;
                    split2(st_reg[2 * i], st_reg[2 * i + 1], &hi, &lo);
                    *reinterpret_cast<uint32_t*>(p + 4 * i) = hi;
                    *reinterpret_cast<uint32_t*>(p + C::kPlane + 4 * i) = lo;
                }
                if (VW > DH && st_part == 3) {  // channels DH .. VW-1: the ones column, then zeros (16-B aligned)
                    constexpr uint32_t kOne = 0x3F80u;  // bf16 1.0 in the low half: channel DH
#pragma unroll
                    for (int i = 0; i < (VW - DH) / 8; ++i) {
                        *reinterpret_cast<u32x4*>(dst + DH * 2 + 16 * i) = (u32x4){i == 0 ? kOne : 0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4*>(dst + C::kPlane + DH * 2 + 16 * i) = (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
        }
    };

    // The first key tile's gather is issued BEFORE the query prologue: its two dependent round trips (token index, then
    // the rows) overlap those of the queries instead of following them -- these workgroups live for a few microseconds,
    // most of it memory latency.
    int32_t tok_next = n_kt > 1 ? load_tok(1) : 0;
    {
        const int32_t tok0 = load_tok(0);
        AKEEP(tok0);
        ASTAMP(1);  // token indices of the first key tile
        stage_load(tok0);
    }

    // ---------------------------------------------------------------- this wave's (tile, head) units
    const int n_qt_here = min(QT, n_kt - item.y * QT);  // query tiles of the item that exist
    bf16x8 q_hi[UW][2][KS], q_lo[UW][2][KS];
    f32x4 o_acc[UW][2][NB];
    float m_run[UW][2], l_run[UW][2];
    int32_t token[UW][2];
    int q0[UW], hh_of[UW];
    bool active[UW], two[UW];
    uint32_t drop_row[UW][2];  // dropout: hash state of this lane's query pair (attn_dropout.hpp), fixed over the key loop
#pragma unroll
    for (int un = 0; un < UW; ++un) {
        const int unit = wave + 4 * un;
        const int qt = C::kNarrow ? 0 : unit;
        hh_of[un] = C::kNarrow ? unit : 0;
        q0[un] = (item.y * QT + qt) * 32;
        active[un] = qt < n_qt_here;          // wave-uniform
        two[un] = n - q0[un] > 16;            // the tile's second 16-query group exists (wave-uniform)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int qi = q0[un] + 16 * j + c16;
            if constexpr (DROPOUT) drop_row[un][j] = dropout_row_state(dropout_head_state(drop, item.x, h0 + hh_of[un]), qi);
            token[un][j] = -1;
            m_run[un][j] = -INFINITY;
            l_run[un][j] = 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) o_acc[un][j][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // a wave without this query group (idle wave of a small window, tile with <= 16 queries) skips the whole
            // normalise / split prologue: wave-uniform, and those fragments are never multiplied
            if (!active[un] || (j == 1 && !two[un])) {
#pragma unroll
                for (int s = 0; s < KS; ++s) q_hi[un][j][s] = q_lo[un][j][s] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
                continue;
            }
            float x[KS][8];
            float ss = 0.f;
            const bool have = qi < n;
            if (have) token[un][j] = tok[start + qi];
            const float* row = q + (int64_t)(have ? token[un][j] : 0) * ldq + (h0 + hh_of[un]) * DH;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c0 = 32 * s + 8 * g + 2 * i;
                    f32x2 xv = {0.f, 0.f};
                    if (have && c0 < DH) xv = *reinterpret_cast<const f32x2*>(row + c0);
                    x[s][2 * i] = xv[0];
                    x[s][2 * i + 1] = xv[1];
                    ss = fmaf(xv[0], xv[0], fmaf(xv[1], xv[1], ss));
                }
            ss += __shfl_xor(ss, 16, SEG3D_WAVE);
            ss += __shfl_xor(ss, 32, SEG3D_WAVE);
            const float r = qscale * inv_norm(ss);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int i = 0; i < 8; ++i) x[s][i] *= r;
                split_frag(x[s], &q_hi[un][j][s], &q_lo[un][j][s]);
            }
        }
    }

    // ---------------------------------------------------------------- fragment reads of a staged tile
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    auto read_k = [&](const char* base, int hh, int u, int s, bf16x8* hi, bf16x8* lo) {
        const int c0 = 32 * s + 8 * g;
        if (c0 < DHS) {
            const char* p = base + (u * 16 + c16) * KRS + (hh * DHS + c0) * 2;
            *hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
            *lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + C::kPlane));
        } else {
            *hi = *lo = __builtin_bit_cast(bf16x8, zero4);
        }
    };
    // V^T fragment of d-block b: A operand, lane (d = 16 b + c16, key slots 8 g .. 8 g + 7 = keys 4g..4g+3, 16+4g..16+4g+3).
    // Transposed out of the row-major image by two ds_read_b64_tr_b16 per plane: lane 4 q' + p' of a 16-lane group
    // supplies the address of key row q', channels 4 p' .. 4 p' + 3 and receives channel c16 of the four rows.
    // Every lane takes part (EXEC is all ones here: the surrounding branches are wave-uniform).
    auto read_vt = [&](const char* base, int hh, int b, bf16x8* hi, bf16x8* lo) {
        const int qq = c16 >> 2, pp = c16 & 3;
        const char* p0 = base + 32 * KRS + (4 * g + qq) * VRS + (hh * VW + 16 * b + 4 * pp) * 2;
        const char* p1 = p0 + 16 * VRS;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + C::kPlane));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1 + C::kPlane));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        *hi = __builtin_bit_cast(bf16x8, (s16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
        *lo = __builtin_bit_cast(bf16x8, (s16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
    };

    // ---------------------------------------------------------------- one key tile for one unit
    auto tile_step = [&](auto fixed_tag, int un, int t, const char* base) {
        constexpr bool FIXED = decltype(fixed_tag)::value;
        const bool last = t + 1 == n_kt;
        const int hh = hh_of[un];
        bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[NB], v_lo[NB];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) read_k(base, hh, u, s, &k_hi[u][s], &k_lo[u][s]);
#pragma unroll
        for (int b = 0; b < NB; ++b) read_vt(base, hh, b, &v_hi[b], &v_lo[b]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two[un]) break;
            float sc[8];
            const float init = FIXED ? -qscale : 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 acc = {init, init, init, init};
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = mfma3(k_hi[u][s], k_lo[u][s], q_hi[un][j][s], q_lo[un][j][s], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[u * 4 + r] = acc[r];
            }
            float alpha = 1.0f;
            if constexpr (FIXED) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i]);
                if (last) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (t * 32 + (i >> 2) * 16 + g * 4 + (i & 3) >= n) sc[i] = 0.f;
                }
            } else {
                if (last) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (t * 32 + (i >> 2) * 16 + g * 4 + (i & 3) >= n) sc[i] = -INFINITY;
                }
                float tmax = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, SEG3D_WAVE));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, SEG3D_WAVE));
                const float m_new = fmaxf(m_run[un][j], tmax);
                alpha = __builtin_amdgcn_exp2f(m_run[un][j] - m_new);
                m_run[un][j] = m_new;
#pragma unroll
                for (int i = 0; i < 8; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_new);
            }
            if constexpr (DROPOUT || !C::kOnes || !FIXED) {
                // row sum on the vector ALUs (per-lane partial: the lanes of a query column are summed once, at the end)
                float ps = ((sc[0] + sc[1]) + (sc[2] + sc[3])) + ((sc[4] + sc[5]) + (sc[6] + sc[7]));
                l_run[un][j] = fmaf(l_run[un][j], alpha, ps);
            }
            if constexpr (DROPOUT) {
                // keys 4g .. 4g+3 of each 16-key half = two 2 x 2 blocks shared with lane c16 ^ 1 (same query pair):
                // the even lane hashes the first, the odd lane the second, one DPP swap (dropout_pair_bits)
                const int qi = q0[un] + 16 * j + c16;
                const bool odd = c16 & 1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kj = t * 32 + u * 16 + 4 * g;
                    uint32_t bits[2];
                    dropout_pair_bits(dropout_block_bits(drop_row[un][j], dropout_key_term(kj + (odd ? 2 : 0))), odd, &bits[0], &bits[1]);
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        if (dropout_dropped(drop, bits[r2], qi, kj + 2 * r2)) sc[u * 4 + 2 * r2] = 0.f;
                        if (dropout_dropped(drop, bits[r2], qi, kj + 2 * r2 + 1)) sc[u * 4 + 2 * r2 + 1] = 0.f;
                    }
                }
            }
            bf16x8 p_hi, p_lo;
            split_frag(sc, &p_hi, &p_lo);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                f32x4 acc = o_acc[un][j][b];
                if constexpr (!FIXED) acc = acc * alpha;
                o_acc[un][j][b] = mfma3(v_hi[b], v_lo[b], p_hi, p_lo, acc);
            }
        }
    };

    // ---------------------------------------------------------------- epilogue of a unit
    auto finish = [&](bool fixed, int un) {
        const int h = h0 + hh_of[un];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two[un]) break;
            float l;
            if (C::kOnes && !DROPOUT && fixed) {  // the ones column: d = DH lives in block DH / 16, lane group (DH % 16) / 4
                constexpr int b1 = DH / 16, g1 = (DH % 16) / 4, r1 = DH % 4;
                l = __shfl(o_acc[un][j][b1][r1], c16 + 16 * g1, SEG3D_WAVE);
            } else {
                l = l_run[un][j];
                l += __shfl_xor(l, 16, SEG3D_WAVE);
                l += __shfl_xor(l, 32, SEG3D_WAVE);
            }
            if (token[un][j] < 0) continue;
            const float inv = (DROPOUT ? drop.inv_keep : 1.0f) * __builtin_amdgcn_rcpf(l);  // 1 ulp; products carry 2^-16
            float* op = out + (int64_t)token[un][j] * (heads * DH) + h * DH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int d = 16 * b + 4 * g;
                const f32x4 o = o_acc[un][j][b] * inv;
                if (DH % 4 == 0) {
                    if (d < DH) *reinterpret_cast<f32x4*>(op + d) = o;
                } else {
                    if (d + 1 < DH) *reinterpret_cast<f32x2*>(op + d) = (f32x2){o[0], o[1]};
                    if (d + 3 < DH) *reinterpret_cast<f32x2*>(op + d + 2) = (f32x2){o[2], o[3]};
                }
            }
            const float mx = fixed ? qscale : m_run[un][j];
            if (lse && g == 0) lse[(int64_t)token[un][j] * heads + h] = (mx + __builtin_amdgcn_logf(l)) * kLn2;
        }
    };

    // ---------------------------------------------------------------- main loop over the window's key tiles
    auto run = [&](auto fixed_tag) {
        ASTAMP(2);  // query prologue (token, row gather, normalise, split)
        stage_store(0);
        ASTAMP(3);  // wait for the first key tile's rows + convert + LDS store
        __syncthreads();
        ASTAMP(4);  // barriers
        for (int t = 0; t < n_kt; ++t) {
            const bool more = t + 1 < n_kt;
            const int buf = C::NBUF == 2 ? (t & 1) : 0;
            if (more) {
                stage_load(tok_next);  // in flight while this tile is multiplied
                if (t + 2 < n_kt) tok_next = load_tok(t + 2);
            }
#pragma unroll
            for (int un = 0; un < UW; ++un)
                if (active[un]) tile_step(fixed_tag, un, t, lds + buf * C::kTile);
            ASTAMP(5);  // tile compute (LDS fragment reads, MFMAs, softmax)
            if (C::NBUF == 1) __syncthreads();  // everyone is done with the only buffer
            ASTAMP(4);
            if (more) stage_store(C::NBUF == 2 ? (buf ^ 1) : 0);
            ASTAMP(3);
            __syncthreads();
            ASTAMP(4);
        }
    };
    if (fixed_max) run(std::true_type{});
    else run(std::false_type{});
#pragma unroll
    for (int un = 0; un < UW; ++un)
        if (active[un]) finish(fixed_max, un);
#ifdef SEG3D_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(6);  // epilogue (normalise, stores)
    if (g_attn_stamp_buf && lane == 0) {
        unsigned long long* o = g_attn_stamp_buf + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        st_acc[7] = st_last - st_begin;
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
}

template <int DH>
int launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
           const int32_t* win_start, const int32_t* win_count, const int2* tile_item, int n_tiles, const int2* chunk_item,
           int n_chunks, int heads, const float* tau, float tau_min, float* out, float* lse, const DropoutParams& drop,
           hipStream_t st) {
    using C = Cfg<DH>;
    const int2* items = C::kNarrow ? tile_item : chunk_item;
    const int n_items = C::kNarrow ? n_tiles : n_chunks;
    const dim3 grid((unsigned)n_items, (unsigned)(heads / C::HG));
    if (drop.threshold)
        hipLaunchKernelGGL((attn_fused_fwd<DH, true>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, tok, win_start, win_count,
                           items, heads, tau, tau_min, out, lse, drop);
    else
        hipLaunchKernelGGL((attn_fused_fwd<DH, false>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, tok, win_start, win_count,
                           items, heads, tau, tau_min, out, lse, drop);
    SEG3D_CHECK_LAUNCH();
    return SEG3D_OK;
}

}  // namespace

// the narrow configuration takes 4 heads per workgroup: other head counts stay on the two-launch kernels
bool attn_fused_supported(int heads, int dh) {
    if (dh == 6 || dh == 12) return heads % 4 == 0;
    return dh == 24 || dh == 48;
}

int attn_fused_fwd_launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
                          const int32_t* win_start, const int32_t* win_count, const int32_t* tile_item, int n_tiles,
                          const int32_t* chunk_item, int n_chunks, int heads, int dh, const float* tau, float tau_min,
                          float* out, float* lse, float dropout_p, uint64_t seed, hipStream_t st) {
    const DropoutParams drop = make_dropout(dropout_p, seed);
    const int2* ti = reinterpret_cast<const int2*>(tile_item);
    const int2* ci = reinterpret_cast<const int2*>(chunk_item);
    switch (dh) {
        case 6: return launch<6>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 12: return launch<12>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 24: return launch<24>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 48: return launch<48>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        default: return SEG3D_EINVAL;
    }
}

// a19-a21 forward, fused: ragged sparse-window cosine attention in ONE launch per layer.
// Reference: flat2window -> CosineMultiheadAttention -> window2flat (swformer_utils.py:34-85,
// point_transformer_layer.py:233-258, cosine_msa.py:115-177).
//
// What the earlier forward did in two launches (attn_prepare_fwd: gather + L2-normalise + bf16 hi/lo split + transposed
// copies through HBM; attn_core_fwd_lds: the MFMA core) happens here inside the core's own staging step:
//   * PERSISTENT workgroups (4 waves): the grid is one round of resident workgroups, each walks its share of the work
//     items (item = blockIdx.x + j * gridDim.x) in a software-pipelined loop.  In-kernel stamps of the one-item-per-
//     workgroup form (tools/probes/attn_stamps.py) showed where a workgroup's life went: 10-18 % fetching its item record
//     and window geometry, 25-40 % in the query prologue (token index -> scattered 8-B row gathers -> normalise), 8-21 %
//     waiting for key rows, and only 19-28 % multiplying -- four dependent memory round trips per item with nothing to
//     overlap them.  Here the item descriptors of a workgroup are resolved ONCE into LDS, the token indices of item j + 1
//     are fetched while item j's first key tile is multiplied, and its query / first-key rows are requested before item
//     j's epilogue: the chain of item j + 1 runs under the arithmetic of item j;
//   * the queries take the same road as the keys: all 256 threads gather whole 16-B pieces of the query rows (4 threads
//     per row), normalise, scale by log2e / tau, split and park them in an LDS image; a wave pulls its B fragments out
//     of it with ds_read_b128 (the scattered 8-B fragment-shaped global loads of the earlier prologue are gone);
//   * a workgroup owns the queries of one work item and walks the window's 32-key tiles.  Per key tile all 256
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
#include <cstdlib>
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


// LDS the persistent forward needs beyond the key / value tiles: the query image (two planes) + the tokens of its rows
// + the workgroup's resolved item descriptors (a ring of kDescRing)
constexpr int kDescRing = 256;
#ifndef SEG3D_ATTN_DH6_WAVES
#define SEG3D_ATTN_DH6_WAVES 4
#endif
template <int DH>
struct FwdGeo {
    using C = Cfg<DH>;
    static constexpr int QROWS = C::QT * 32;                       // stationary (query) rows of an item
    static constexpr int QRS = C::HG * C::DHS * 2 + (C::kNarrow ? 16 : (DH == 24 ? 16 : 16));  // bytes per row per plane
    static constexpr int kQPlane = QROWS * QRS;
    static constexpr int QP = C::kNarrow ? 1 : 2;                  // staging passes over the query rows (64 rows x 4 threads)
    static constexpr int NBUF = C::NBUF;  // (double-buffering the narrow configurations too lost: 77 -> 87 us at dh 6)
    static constexpr int kBytes = NBUF * C::kTile + 2 * kQPlane + QROWS * 4 + kDescRing * 16;
    // waves per SIMD = resident workgroups per CU the register allocator leaves room for (spill-free points)
    static constexpr int waves(bool dropout) { return DH == 48 ? 2 : (DH == 6 && !dropout ? SEG3D_ATTN_DH6_WAVES : 3); }
};

template <int DH, bool DROPOUT>
__global__ __launch_bounds__(256, FwdGeo<DH>::waves(DROPOUT)) void attn_fused_fwd(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ldq, int ldk, int ldv,
    const int32_t* __restrict__ tok, const int32_t* __restrict__ win_start, const int32_t* __restrict__ win_count,
    const int4* __restrict__ items, int n_items, int heads, const float* __restrict__ tau, float tau_min,
    float* __restrict__ out, float* __restrict__ lse, DropoutParams drop, int xcd_groups, int xcd_block) {
    using C = Cfg<DH>;
    using F = FwdGeo<DH>;
    constexpr int HG = C::HG, QT = C::QT, DHS = C::DHS, KS = C::KS, VW = C::VW, NB = C::NB;
    constexpr int KRS = C::KRS, VRS = C::VRS, CT = C::CT, QRS = F::QRS, QP = F::QP;
    static_assert(C::UW == 1, "one (tile, head) unit per wave");
    // the spare channel of the score product (attn_common.hpp: kSpareOne / kSpareMask): channel DHS = fragment SP_KS,
    // lane group SP_G, element 0 -- keys past the window's end are switched off inside the score MFMAs
    constexpr int SP_KS = DHS / 32, SP_G = (DHS % 32) / 8;
    static_assert(KS * 32 > DHS, "the score product needs a spare K channel");
    __shared__ __attribute__((aligned(16))) char lds[F::kBytes];
    constexpr int NBUF = F::NBUF;
    char* const q_lds = lds + NBUF * C::kTile;
    int32_t* const qtok_lds = reinterpret_cast<int32_t*>(q_lds + 2 * F::kQPlane);
    int4* const desc = reinterpret_cast<int4*>(q_lds + 2 * F::kQPlane + F::QROWS * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    if (tid < NBUF * 2 * 4)  // the zero block behind each plane of each key-tile buffer (never written again)
        *reinterpret_cast<uint32_t*>(lds + (tid >> 3) * C::kTile + ((tid >> 2) & 1) * C::kPlane + C::kPlaneData + (tid & 3) * 4) = 0u;
#ifdef SEG3D_ATTN_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    const int hgn = heads / HG;                 // head groups per item
    // Work units = (item, head group), head group fastest.  A token row holds all heads, and a head's slice of it is a
    // fraction of a cache line: the workgroups that read the same rows must share an L2, or every XCD pulls the same lines
    // over the fabric.  Workgroups are dealt round-robin over the XCDs (blockIdx % xcd_groups labels the workgroups that
    // share one; placement is a speed matter only), so group x takes the items = x (mod xcd_groups) and walks them with
    // its own workgroups side by side: at any moment one XCD works on a handful of windows, all heads of each.
    const int XG = xcd_groups;                  // 1 = flat order
    const int gx = (int)blockIdx.x % XG, gs = (int)blockIdx.x / XG;
    const int S = ((int)gridDim.x - gx + XG - 1) / XG;       // workgroups of my group
    const int units_x = xcd_block_count(n_items, gx, XG, xcd_block) * hgn;  // units of my group: u = s + j * S
    const int J = units_x > gs ? (units_x - gs + S - 1) / S : 0;
    auto unit_of = [&](int j, int* item, int* hg) {
        const int u = gs + j * S;
        *item = xcd_block_item(u / hgn, gx, XG, xcd_block);  // blocks of consecutive items (one window's tiles) per XCD
        *hg = u % hgn;
    };
    const float qscale = kLog2e / fmaxf(tau[0], tau_min);
    // every score is <= qscale; p = exp2(s - qscale) >= 2^(-2 qscale) must stay a normal float
    const bool fixed_max = qscale <= 40.0f;

    // resolve descriptors j0 .. j0 + count - 1 of this workgroup into the ring: {first token slot, tokens, tile / chunk, window}
    auto fill_desc = [&](int j0, int count) {
        if (tid < count && j0 + tid < J) {
            int item_i, hg_i;
            unit_of(j0 + tid, &item_i, &hg_i);
            const int4 it = items[item_i];  // {window, tile / chunk, first token slot, tokens}: no dependent win_start / win_count reads
            desc[(j0 + tid) & (kDescRing - 1)] = make_int4(it.z, it.w, it.y, it.x);
        }
    };

    // ---------------------------------------------------------------- staging role of this thread: key / value tiles
    const int st_which = tid >> 7;            // 0: K rows, 1: V rows
    const int st_key = (tid & 127) >> 2, st_part = tid & 3;
    const float* st_src = st_which == 0 ? k : v;
    const int st_ld = st_which == 0 ? ldk : ldv;
    // first channel of this thread's share, relative to the head group's first channel
    const int st_col0 = C::kNarrow ? (C::HPT * st_part) * DH : st_part * CT;
    float st_reg[CT];
    auto load_part = [&](const float* row, float* dst) {
        // widest aligned pieces: the share starts at a multiple of CT floats past a 16-B aligned head-group base
        if constexpr ((CT * 4) % 16 == 0 && (C::kNarrow ? (C::HPT * DH * 4) % 16 == 0 : (DH * 4) % 16 == 0)) {
#pragma unroll
            for (int i = 0; i < CT / 4; ++i) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * i);
                dst[4 * i] = x[0]; dst[4 * i + 1] = x[1]; dst[4 * i + 2] = x[2]; dst[4 * i + 3] = x[3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CT / 2; ++i) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(row + 2 * i);
                dst[2 * i] = x[0]; dst[2 * i + 1] = x[1];
            }
        }
    };
    // normalised (x scale) + split image row piece: [row][head][DHS] hi plane, lo plane `plane` bytes behind
    auto store_norm = [&](const float* reg, float scale, char* row_dst, int plane) {
        if constexpr (C::kNarrow) {
#pragma unroll
            for (int hh = 0; hh < C::HPT; ++hh) {
                float ss = 0.f;
#pragma unroll
                for (int d = 0; d < DH; ++d) ss = fmaf(reg[hh * DH + d], reg[hh * DH + d], ss);
                const float r = scale * inv_norm(ss);
                uint32_t hi[DHS / 2], lo[DHS / 2];
#pragma unroll
                for (int i = 0; i < DHS / 2; ++i) {
                    const float a = 2 * i < DH ? reg[hh * DH + 2 * i] * r : 0.f;
                    const float b = 2 * i + 1 < DH ? reg[hh * DH + 2 * i + 1] * r : 0.f;
                    split2(a, b, &hi[i], &lo[i]);
                }
                char* p = row_dst + ((C::HPT * st_part + hh) * DHS) * 2;
#pragma unroll
                for (int i = 0; i < DHS / 8; ++i) {
                    *reinterpret_cast<u32x4*>(p + 16 * i) = (u32x4){hi[4 * i], hi[4 * i + 1], hi[4 * i + 2], hi[4 * i + 3]};
                    *reinterpret_cast<u32x4*>(p + plane + 16 * i) = (u32x4){lo[4 * i], lo[4 * i + 1], lo[4 * i + 2], lo[4 * i + 3]};
                }
            }
        } else {
            float ss = 0.f;
#pragma unroll
            for (int d = 0; d < CT; ++d) ss = fmaf(reg[d], reg[d], ss);
            ss = quad_sum(ss);  // the 4 threads of a row hold a quarter of the head each
            const float r = scale * inv_norm(ss);
            char* p = row_dst + (st_part * CT) * 2;
#pragma unroll
            for (int i = 0; i < CT / 2; ++i) {
                uint32_t hi, lo;
                split2(reg[2 * i] * r, reg[2 * i + 1] * r, &hi, &lo);
                *reinterpret_cast<uint32_t*>(p + 4 * i) = hi;
                *reinterpret_cast<uint32_t*>(p + plane + 4 * i) = lo;
            }
        }
    };
    auto stage_store = [&](int buf) {
        char* base = lds + buf * C::kTile;
        if (st_which == 0) {  // K: L2-normalise per head, split, row-major [key][head][DHS]
            store_norm(st_reg, 1.0f, base + st_key * KRS, C::kPlane);
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

    // ---------------------------------------------------------------- staging role: the item's query rows
    // narrow heads: 32 rows x 4 heads, threads 0 .. 127 (one head of one row each); wide heads: up to 128 rows of one head,
    // two passes of 64 rows x 4 threads
    const bool q_role = C::kNarrow ? tid < 128 : true;
    const int q_row0 = C::kNarrow ? (tid & 127) >> 2 : tid >> 2;  // + 64 per pass
    float q_reg[QP][CT];
    int32_t q_tok[QP];

    // ---------------------------------------------------------------- fragment reads of a staged tile
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // (lanes whose channels lie past the stored head width read the plane's zero block: no branch, no register fill)
    auto read_k = [&](const char* base, int hh, int u, int s, bf16x8* hi, bf16x8* lo) {
        const int c0 = 32 * s + 8 * g;
        const int rel = c0 < DHS ? (u * 16 + c16) * KRS + (hh * DHS + c0) * 2 : C::kPlaneData;
        const char* p = base + rel;
        *hi = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
        *lo = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + C::kPlane));
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

    const uint32_t spare_mask = g == SP_G ? kSpareMask : 0u;
    const uint32_t drop_shift = dropout_lane_shift_query(c16 & 1);  // (queries q0 + 16 jj + c16: q0 is a multiple of 32)
    // this wave's unit inside an item: narrow = head `wave` of the 32-query tile, wide = query tile `wave` of the head
    const int qt = C::kNarrow ? 0 : wave;
    const int hh = C::kNarrow ? wave : 0;

    auto run = [&](auto fixed_tag) {
        constexpr bool FIXED = decltype(fixed_tag)::value;
        if (J == 0) return;  // (workgroup-uniform)
        fill_desc(0, kDescRing);
        __syncthreads();
        ASTAMP(0);  // descriptors of the workgroup (once)
        // ---- item 0: its token indices and rows are fetched here, in the open; every later item's ride under arithmetic
        int4 cur = desc[0];
        int32_t tok_next = 0;
        auto fetch_tokens = [&](const int4& d, int32_t* t0, int32_t* t1, int32_t* tq) {
            const int nn = d.y, ns = d.x;
            *t0 = tok[ns + (st_key < nn ? st_key : nn - 1)];
            *t1 = nn > 32 ? tok[ns + (32 + st_key < nn ? 32 + st_key : nn - 1)] : 0;
#pragma unroll
            for (int p = 0; p < QP; ++p) {
                const int qi = d.z * QT * 32 + q_row0 + 64 * p;
                tq[p] = q_role ? tok[ns + (qi < nn ? qi : nn - 1)] : 0;
            }
        };
        auto request_rows = [&](const int4& d, int h0r, int32_t t0, const int32_t* tq) {
            load_part(st_src + (int64_t)t0 * st_ld + h0r * DH + st_col0, st_reg);
#pragma unroll
            for (int p = 0; p < QP; ++p) {
                q_tok[p] = tq[p];
                if (q_role && (p == 0 || d.y - d.z * QT * 32 > 64))
                    load_part(q + (int64_t)tq[p] * ldq + h0r * DH + st_col0, q_reg[p]);
            }
        };
        {
            int32_t t0, tq[QP];
            fetch_tokens(cur, &t0, &tok_next, tq);
            int item0, hg0;
            unit_of(0, &item0, &hg0);
            request_rows(cur, hg0 * HG, t0, tq);
        }
        ASTAMP(1);

        for (int j = 0; j < J; ++j) {
            const int n = cur.y, start = cur.x, win = cur.w;
            int item_j, hg_j;
            unit_of(j, &item_j, &hg_j);
            const int h0 = hg_j * HG;
            const int n_kt = (n + 31) >> 5;
            const int q_base = cur.z * QT * 32;        // first query of the item
            const int n_q = min(QT * 32, n - q_base);  // its queries
            if (j > 0 && (j & (kDescRing / 2 - 1)) == 0) fill_desc(j + kDescRing / 2, kDescRing / 2);  // refill the dead half

            // ---- (A) park the query image and the first key tile
#pragma unroll
            for (int p = 0; p < QP; ++p) {
                if (q_role && (p == 0 || n_q > 64)) {
                    const int row = q_row0 + 64 * p;
                    store_norm(q_reg[p], qscale, q_lds + row * QRS, F::kQPlane);
                    if (st_part == 0) qtok_lds[row] = q_base + row < n ? q_tok[p] : -1;
                }
            }
#ifdef SEG3D_ATTN_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ASTAMP(1);  // (diagnostic build) waiting for prefetched rows
#endif
            stage_store(0);
            ASTAMP(3);
            __syncthreads();
            ASTAMP(4);

            // ---- (B) this wave's query fragments
            const int q0 = q_base + qt * 32;
            const bool active = qt * 32 < n_q;       // wave-uniform
            const bool two = n - q0 > 16;            // the tile's second 16-query group exists (wave-uniform)
            bf16x8 q_hi[2][KS], q_lo[2][KS];
            f32x4 o_acc[2][NB];
            float m_run[2], l_run[2];
            int32_t token[2];
            uint32_t drop_row[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                m_run[jj] = -INFINITY;
                l_run[jj] = 0.f;
                token[jj] = -1;
                drop_row[jj] = 0u;
#pragma unroll
                for (int b = 0; b < NB; ++b) o_acc[jj][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const bool have = active && (jj == 0 || two);
                const int row = qt * 32 + 16 * jj + c16;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int c0 = 32 * s + 8 * g;
                    u32x4 a = zero4, b = zero4;
                    if (have && c0 < DHS) {
                        const char* p = q_lds + row * QRS + (hh * DHS + c0) * 2;
                        a = *reinterpret_cast<const u32x4*>(p);
                        b = *reinterpret_cast<const u32x4*>(p + F::kQPlane);
                    }
                    if (s == SP_KS && g == SP_G) a[0] = kSpareOne;  // the queries' 1.0 in the spare score channel
                    q_hi[jj][s] = __builtin_bit_cast(bf16x8, a);
                    q_lo[jj][s] = __builtin_bit_cast(bf16x8, b);
                }
                if (have) token[jj] = qtok_lds[row];
                if constexpr (DROPOUT) drop_row[jj] = dropout_row_state(dropout_head_state(drop, win, h0 + hh), q0 + 16 * jj + c16);
            }

            // ---- (C) token indices of the next item (first two key tiles, query rows): in flight under this item's tiles.
            // (Fetching them an item earlier and requesting the rows at the start of the last key tile was tried: the row
            // wait fell from 20 % to 11 % of a wave's time, the extra live registers cost more -- 1.57 -> 1.80 ms.)
            const bool more_items = j + 1 < J;
            int4 nxt = cur;
            int32_t n_tok0 = 0, n_tok1 = 0, nq_tok[QP];
#pragma unroll
            for (int p = 0; p < QP; ++p) nq_tok[p] = 0;
            if (more_items) {
                nxt = desc[(j + 1) & (kDescRing - 1)];
                fetch_tokens(nxt, &n_tok0, &n_tok1, nq_tok);
            }

            // ---- (D) the window's key tiles
            auto load_tok = [&](int t) {
                int kk = t * 32 + st_key;
                kk = kk < n ? kk : n - 1;  // clamped rows are finite and masked by p = 0
                return tok[start + kk];
            };
            for (int t = 0; t < n_kt; ++t) {
                const bool more = t + 1 < n_kt;
                const int buf = NBUF == 2 ? (t & 1) : 0;
                if (more) {
                    load_part(st_src + (int64_t)tok_next * st_ld + h0 * DH + st_col0, st_reg);  // in flight while this tile is multiplied
                    if (t + 2 < n_kt) tok_next = load_tok(t + 2);
                }
                if (active) {
                    const char* base = lds + buf * C::kTile;
                    const bool last = t + 1 == n_kt;
                    bf16x8 k_hi[2][KS], k_lo[2][KS], v_hi[NB], v_lo[NB];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int s = 0; s < KS; ++s) read_k(base, hh, u, s, &k_hi[u][s], &k_lo[u][s]);
#pragma unroll
                    for (int b = 0; b < NB; ++b) read_vt(base, hh, b, &v_hi[b], &v_lo[b]);
                    if (last) {  // (wave-uniform) keys past the end: -16384 in the spare channel, against the queries' 1.0
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            u32x4 w = __builtin_bit_cast(u32x4, k_hi[u][SP_KS]);
                            w[0] |= t * 32 + u * 16 + c16 >= n ? spare_mask : 0u;
                            k_hi[u][SP_KS] = __builtin_bit_cast(bf16x8, w);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (jj == 1 && !two) break;
                        float sc[8];
                        const float init = FIXED ? -qscale : 0.f;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            f32x4 acc = {init, init, init, init};
#pragma unroll
                            for (int s = 0; s < KS; ++s) acc = mfma3(k_hi[u][s], k_lo[u][s], q_hi[jj][s], q_lo[jj][s], acc);
#pragma unroll
                            for (int r = 0; r < 4; ++r) sc[u * 4 + r] = acc[r];
                        }
                        float alpha = 1.0f;
                        if constexpr (FIXED) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i]);
                        } else {
                            float tmax = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
                            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, SEG3D_WAVE));
                            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, SEG3D_WAVE));
                            const float m_new = fmaxf(m_run[jj], tmax);
                            alpha = __builtin_amdgcn_exp2f(m_run[jj] - m_new);
                            m_run[jj] = m_new;
#pragma unroll
                            for (int i = 0; i < 8; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_new);
                        }
                        if constexpr (DROPOUT || !C::kOnes || !FIXED) {
                            // row sum on the vector ALUs (per-lane partial: the lanes of a query column are summed once, at the end)
                            float ps = ((sc[0] + sc[1]) + (sc[2] + sc[3])) + ((sc[4] + sc[5]) + (sc[6] + sc[7]));
                            l_run[jj] = fmaf(l_run[jj], alpha, ps);
                        }
                        if constexpr (DROPOUT) {
                            // keys 4g .. 4g+3 of each 16-key half = two 2 x 2 blocks shared with lane c16 ^ 1 (same query pair):
                            // the even lane hashes the first, the odd lane the second, one DPP swap (dropout_pair_bits)
                            const bool odd = c16 & 1;
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int kj = t * 32 + u * 16 + 4 * g;
                                uint32_t bits[2];
                                dropout_pair_bits(dropout_block_bits(drop_row[jj], dropout_key_term(kj + (odd ? 2 : 0))), odd, &bits[0], &bits[1]);
#pragma unroll
                                for (int r2 = 0; r2 < 2; ++r2) {  // the query's parity shifted out once per hash: constant byte selects
                                    const uint32_t adj = bits[r2] >> drop_shift;
                                    if (dropout_dropped_byte(drop, adj, 0)) sc[u * 4 + 2 * r2] = 0.f;
                                    if (dropout_dropped_byte(drop, adj, 1)) sc[u * 4 + 2 * r2 + 1] = 0.f;
                                }
                            }
                        }
                        bf16x8 p_hi, p_lo;
                        split_frag(sc, &p_hi, &p_lo);
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            f32x4 acc = o_acc[jj][b];
                            if constexpr (!FIXED) acc = acc * alpha;
#ifdef SEG3D_ATTN_PV2  // (experiment, not shipped: P . V with two products -- p's low half dropped)
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_lo[b], p_hi, acc, 0, 0, 0);
                            o_acc[jj][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v_hi[b], p_hi, acc, 0, 0, 0);
#else
                            o_acc[jj][b] = mfma3(v_hi[b], v_lo[b], p_hi, p_lo, acc);
#endif
                        }
                    }
                }
                ASTAMP(5);  // tile compute (LDS fragment reads, MFMAs, softmax)
                if (NBUF == 1) __syncthreads();  // everyone is done with the only buffer
                ASTAMP(4);
#ifdef SEG3D_ATTN_STAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ASTAMP(1);
#endif
                if (more) stage_store(NBUF == 2 ? (buf ^ 1) : 0);
                ASTAMP(3);
                __syncthreads();
                ASTAMP(4);
            }

            // ---- (E) the next item's rows: requested before this item's epilogue, needed at the top of the next round
            if (more_items) {
                int item_n, hg_n;
                unit_of(j + 1, &item_n, &hg_n);
                request_rows(nxt, hg_n * HG, n_tok0, nq_tok);
                tok_next = n_tok1;
            }

            // ---- (F) epilogue of this item's unit
            if (active) {
                const int h = h0 + hh;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    if (jj == 1 && !two) break;
                    float l;
                    if constexpr (C::kOnes && !DROPOUT && FIXED) {  // the ones column: d = DH lives in block DH / 16, lane group (DH % 16) / 4
                        constexpr int b1 = DH / 16, g1 = (DH % 16) / 4, r1 = DH % 4;
                        l = __shfl(o_acc[jj][b1][r1], c16 + 16 * g1, SEG3D_WAVE);
                    } else {
                        l = l_run[jj];
                        l += __shfl_xor(l, 16, SEG3D_WAVE);
                        l += __shfl_xor(l, 32, SEG3D_WAVE);
                    }
                    if (token[jj] < 0) continue;
                    const float inv = (DROPOUT ? drop.inv_keep : 1.0f) * __builtin_amdgcn_rcpf(l);  // 1 ulp; products carry 2^-16
                    float* op = out + (int64_t)token[jj] * (heads * DH) + h * DH;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const int d = 16 * b + 4 * g;
                        const f32x4 o = o_acc[jj][b] * inv;
                        if (DH % 4 == 0) {
                            if (d < DH) *reinterpret_cast<f32x4*>(op + d) = o;
                        } else {
                            if (d + 1 < DH) *reinterpret_cast<f32x2*>(op + d) = (f32x2){o[0], o[1]};
                            if (d + 3 < DH) *reinterpret_cast<f32x2*>(op + d + 2) = (f32x2){o[2], o[3]};
                        }
                    }
                    const float mx = FIXED ? qscale : m_run[jj];
                    if (lse && g == 0) lse[(int64_t)token[jj] * heads + h] = (mx + __builtin_amdgcn_logf(l)) * kLn2;
                }
            }
            ASTAMP(6);
            cur = nxt;
        }
    };
    if (fixed_max) run(std::true_type{});
    else run(std::false_type{});
#ifdef SEG3D_ATTN_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(6);
    if (g_attn_stamp_buf && lane == 0) {
        unsigned long long* o = g_attn_stamp_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
        st_acc[7] = st_last - st_begin;
        st_acc[2] = (unsigned long long)J;
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
}

template <int DH>
int launch(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, const int32_t* tok,
           const int32_t* win_start, const int32_t* win_count, const int4* tile_item, int n_tiles, const int4* chunk_item,
           int n_chunks, int heads, const float* tau, float tau_min, float* out, float* lse, const DropoutParams& drop,
           hipStream_t st) {
    using C = Cfg<DH>;
    const int4* items = C::kNarrow ? tile_item : chunk_item;
    const int n_items = C::kNarrow ? n_tiles : n_chunks;
    // one round of resident workgroups (persistent): CUs x workgroups per CU, or fewer when there is less work
    static const int n_cu = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    static const int per_cu_env = getenv("SEG3D_ATTN_WGS_PER_CU") ? atoi(getenv("SEG3D_ATTN_WGS_PER_CU")) : 0;  // A/B
    const int per_cu = per_cu_env > 0 ? per_cu_env : FwdGeo<DH>::waves(drop.threshold != 0);
    const long long total = (long long)n_items * (heads / C::HG);
    static const int xcd_env = getenv("SEG3D_ATTN_XCD") ? atoi(getenv("SEG3D_ATTN_XCD")) : 8;  // A/B: 1 = flat order
    const int xg = xcd_env > 0 ? xcd_env : 8;
    const int xb = xg == 1 ? 1 : xcd_block_items(C::kNarrow, n_items);
    long long wgs = total < (long long)n_cu * per_cu ? total : (long long)n_cu * per_cu;
    wgs = (wgs + xg - 1) / xg * xg;  // whole groups (a workgroup without units returns at once)
    const dim3 grid((unsigned)wgs);
    if (drop.threshold)
        hipLaunchKernelGGL((attn_fused_fwd<DH, true>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, tok, win_start, win_count,
                           items, n_items, heads, tau, tau_min, out, lse, drop, xg, xb);
    else
        hipLaunchKernelGGL((attn_fused_fwd<DH, false>), grid, dim3(256), 0, st, q, k, v, ldq, ldk, ldv, tok, win_start, win_count,
                           items, n_items, heads, tau, tau_min, out, lse, drop, xg, xb);
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
    const int4* ti = reinterpret_cast<const int4*>(tile_item);
    const int4* ci = reinterpret_cast<const int4*>(chunk_item);
    switch (dh) {
        case 6: return launch<6>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 12: return launch<12>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 24: return launch<24>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        case 48: return launch<48>(q, k, v, ldq, ldk, ldv, tok, win_start, win_count, ti, n_tiles, ci, n_chunks, heads, tau, tau_min, out, lse, drop, st);
        default: return SEG3D_EINVAL;
    }
}

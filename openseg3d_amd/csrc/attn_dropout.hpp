// Attention-probability dropout (cosine_msa.py:172-174: F.dropout on the softmax output, p = attn_drop = 0.1 through
// point_transformer_layer.py:222-231) as a counter-based mask: element (window, head, query i, key j) is dropped iff one
// byte of a 32-bit hash of (seed, window, head, i >> 1, j >> 1) is below a threshold.  The mask is a pure function of its
// coordinates, so the forward and both backward passes regenerate it independently (no mask tensor, no RNG state),
// whatever their tiling.  One hash covers a 2 x 2 block of (query, key) pairs, i.e. two of the elements a lane holds in
// either the S^T = K.Q^T (forward, dQ pass) or the S = Q.K^T (dK / dV pass) orientation; the other two belong to the
// neighbouring lane (c16 ^ 1), so a lane pair computes each hash once and swaps it by DPP (dropout_pair_bits).
// Drop probability = threshold / 256 (the nearest 8-bit value to p: 26 / 256 = 0.1016 for p = 0.1); kept elements are
// scaled by 256 / (256 - threshold), the inverse of THAT probability, so E[mask] = 1 exactly as in F.dropout.
//
// hash = mix1( R(query pair) ^ key_pair * 0x9E3779 ),  R = mix(mix(seed, window, head) ^ query_pair):
// the two-round mix (24-bit multiplies / xor-shifts: v_mul_u32_u24 runs at full rate on CDNA, 32-bit multiplies at a
// quarter) is spent once per query pair -- hoisted out of the key loop where the query is lane-fixed, staged in LDS per
// query tile where queries stream -- and each 2 x 2 block costs one multiply, one xor and ONE mixing round (9 vector
// instructions instead of 15; together with the pair swap the mask costs 26 instructions per 8 elements instead of 60).
// Statistics (drop rate, byte / neighbour correlations within 2 sigma of an ideal generator, chi-square of the byte
// histogram) measured on the host restatement and on the device: tests/test_gpu_attention.py.
#pragma once
#include <stdint.h>

struct DropoutParams {
    uint32_t threshold;  // 0 = no dropout
    uint32_t seed_lo, seed_hi;
    float inv_keep;      // 256 / (256 - threshold)
};

static inline DropoutParams make_dropout(float p, uint64_t seed) {
    DropoutParams d;
    int t = p > 0.f ? (int)(p * 256.0f + 0.5f) : 0;
    if (p > 0.f && t < 1) t = 1;
    if (t > 255) t = 255;
    d.threshold = (uint32_t)t;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    d.inv_keep = 256.0f / (256.0f - (float)t);
    return d;
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t dropout_mix(uint32_t x) {
    x ^= x >> 15;
    x = (x & 0xFFFFFFu) * 0x9E3779u + (x >> 24) * 0x85EBCAu;  // both factors fit 24 bits: v_mul_u32_u24 / v_mad_u32_u24
    x ^= x >> 13;
    x = (x & 0xFFFFFFu) * 0xC2B2AFu + (x >> 24) * 0x27D4EBu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t dropout_mix1(uint32_t x) {
    x ^= x >> 15;
    x = (x & 0xFFFFFFu) * 0xC2B2AFu + (x >> 24) * 0x27D4EBu;
    x ^= x >> 16;
    return x;
}

// per (window, head); per query pair (hoisted / staged); per key pair
__device__ __forceinline__ uint32_t dropout_head_state(const DropoutParams& d, int window, int head) {
    const uint32_t x = d.seed_lo ^ ((uint32_t)window * 0x9E3779B1u);
    return dropout_mix(x + (uint32_t)head * 0x7F4A7C15u + d.seed_hi);
}
__device__ __forceinline__ uint32_t dropout_row_state(uint32_t head_state, int qi) {
    return dropout_mix(head_state ^ (uint32_t)(qi >> 1));
}
__device__ __forceinline__ uint32_t dropout_key_term(int kj) { return ((uint32_t)(kj >> 1) & 0xFFFFFFu) * 0x9E3779u; }

// 4 mask bytes of the 2 x 2 block that holds (qi, kj): byte (qi & 1) * 2 + (kj & 1)
__device__ __forceinline__ uint32_t dropout_block_bits(uint32_t row_state, uint32_t key_term) {
    return dropout_mix1(row_state ^ key_term);
}
__device__ __forceinline__ uint32_t dropout_bits(const DropoutParams& d, int window, int head, int qi, int kj) {
    return dropout_block_bits(dropout_row_state(dropout_head_state(d, window, head), qi), dropout_key_term(kj));
}

// A lane holds elements of blocks 0 and 1 (two per block); lane c16 ^ 1 holds the other halves of the same two blocks.
// The even lane hashes block 0, the odd lane block 1, one DPP swap (quad_perm [1,0,3,2]) hands each the other's result:
// mine = this lane's hash (of block `c16 & 1`), returns the bits of block 0 and block 1 in b0 / b1.
__device__ __forceinline__ void dropout_pair_bits(uint32_t mine, bool odd, uint32_t* b0, uint32_t* b1) {
    const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);
    *b0 = odd ? other : mine;
    *b1 = odd ? mine : other;
}

__device__ __forceinline__ bool dropout_dropped(const DropoutParams& d, uint32_t bits, int qi, int kj) {
    return ((bits >> (8 * ((qi & 1) * 2 + (kj & 1)))) & 0xFFu) < d.threshold;
}

// Lane-view of a block's four mask bytes.  A lane of the attention kernels is fixed to ONE query (S^T = K.Q^T orientation:
// forward, dQ pass) or ONE key (S = Q.K^T orientation: dK / dV pass); that coordinate's parity selects a byte PAIR of every
// block's hash, the same for all of the lane's elements.  Shifting the pair down once per hash leaves byte indices that are
// compile-time constants per element (v_cmp on a byte select instead of a variable v_bfe_u32 + v_cmp per element):
//   query-fixed lane: adj = bits >> 16 * (qi & 1), key kj of the block reads byte (kj & 1)
//   key-fixed lane:   adj = bits >> 8 * (kj & 1),  query qi of the block reads byte 2 * (qi & 1)
__device__ __forceinline__ uint32_t dropout_lane_shift_query(bool query_odd) { return query_odd ? 16u : 0u; }
__device__ __forceinline__ uint32_t dropout_lane_shift_key(bool key_odd) { return key_odd ? 8u : 0u; }
__device__ __forceinline__ bool dropout_dropped_byte(const DropoutParams& d, uint32_t adj, int byte) {
    return ((adj >> (8 * byte)) & 0xFFu) < d.threshold;
}

// the factor F.dropout multiplies the probability with: 0 or 1 / keep_prob
__device__ __forceinline__ float dropout_factor(const DropoutParams& d, uint32_t bits, int qi, int kj) {
    return dropout_dropped(d, bits, qi, kj) ? 0.0f : d.inv_keep;
}
#endif

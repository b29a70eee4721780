// Attention-probability dropout (cosine_msa.py:172-174: F.dropout on the softmax output, p = attn_drop = 0.1 through
// point_transformer_layer.py:222-231) as a counter-based mask: element (window, head, query i, key j) is dropped iff one
// byte of hash(seed, window, head, i >> 1, j >> 1) is below a threshold.  The mask is a pure function of its coordinates,
// so the forward and both backward passes regenerate it independently (no mask tensor, no RNG state), whatever their tiling.
// One 32-bit hash covers a 2 x 2 block of (query, key) pairs, i.e. two of the elements a lane holds in either the
// S^T = K.Q^T (forward, dQ pass) or the S = Q.K^T (dK / dV pass) orientation.
// Drop probability = threshold / 256 (the nearest 8-bit value to p: 26 / 256 = 0.1016 for p = 0.1); kept elements are
// scaled by 256 / (256 - threshold), the inverse of THAT probability, so E[mask] = 1 exactly as in F.dropout.
// The hash is two rounds of 24-bit multiply / xor-shift (v_mul_u32_u24 runs at full rate on CDNA; 32-bit multiplies at a
// quarter): avalanche measured on the coordinates used here in tests/test_gpu_parity.py::test_attention_dropout_statistics.
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

// 4 mask bytes of the 2 x 2 block that holds (qi, kj): byte (qi & 1) * 2 + (kj & 1)
__device__ __forceinline__ uint32_t dropout_bits(const DropoutParams& d, int window, int head, int qi, int kj) {
    uint32_t x = d.seed_lo ^ ((uint32_t)window * 0x9E3779B1u);
    x = dropout_mix(x + (uint32_t)head * 0x7F4A7C15u + d.seed_hi);
    x = dropout_mix(x ^ (((uint32_t)(qi >> 1) << 16) | (uint32_t)(kj >> 1)));
    return x;
}

__device__ __forceinline__ bool dropout_dropped(const DropoutParams& d, uint32_t bits, int qi, int kj) {
    return ((bits >> (8 * ((qi & 1) * 2 + (kj & 1)))) & 0xFFu) < d.threshold;
}

// the factor F.dropout multiplies the probability with: 0 or 1 / keep_prob
__device__ __forceinline__ float dropout_factor(const DropoutParams& d, uint32_t bits, int qi, int kj) {
    return dropout_dropped(d, bits, qi, kj) ? 0.0f : d.inv_keep;
}
#endif

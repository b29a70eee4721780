// Shared device/host helpers for libseg3d_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/seg3d_hip.h"

#define SEG3D_WAVE 64

// The failing hipError_t is kept (per thread) for seg3d_last_error(): a bare SEG3D_ELAUNCH says nothing about why.
void seg3d_note_hip_error(int code, const char* file, int line);

#define SEG3D_CHECK_LAUNCH()                                          \
    do {                                                              \
        const hipError_t seg3d_e_ = hipGetLastError();                \
        if (seg3d_e_ != hipSuccess) {                                 \
            seg3d_note_hip_error((int)seg3d_e_, __FILE__, __LINE__);  \
            return SEG3D_ELAUNCH;                                     \
        }                                                             \
    } while (0)

// same for runtime calls that return their hipError_t (memsets, rocPRIM launches)
#define SEG3D_CHECK_HIP(expr)                                         \
    do {                                                              \
        const hipError_t seg3d_e_ = (expr);                           \
        if (seg3d_e_ != hipSuccess) {                                 \
            seg3d_note_hip_error((int)seg3d_e_, __FILE__, __LINE__);  \
            return SEG3D_ELAUNCH;                                     \
        }                                                             \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve a sub-buffer out of a caller-provided workspace (256-B aligned pieces)
struct WsCarver {
    char* base;
    size_t off;
    explicit WsCarver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align_up(count * sizeof(T), 256);
        return p;
    }
};

// ---------------------------------------------------------------- coordinate hash
// Open addressing, linear probing.  keys: 64-bit linear site index, EMPTY = all ones.
// Layout in the table buffer: [cap] uint64 keys, then [cap] int32 values.
#define SEG3D_HASH_EMPTY 0xFFFFFFFFFFFFFFFFull

static inline uint64_t hash_capacity(int64_t m) {
    uint64_t cap = 1024;
    while (cap < (uint64_t)(2 * m + 2)) cap <<= 1;
    return cap;
}

struct HashView {
    unsigned long long* keys;
    int32_t* vals;
    uint64_t mask;
};

static inline HashView hash_view(void* table, uint64_t cap) {
    HashView h;
    h.keys = reinterpret_cast<unsigned long long*>(table);
    h.vals = reinterpret_cast<int32_t*>(h.keys + cap);
    h.mask = cap - 1;
    return h;
}

__device__ __forceinline__ uint64_t hash_mix(uint64_t k) {
    k *= 0x9E3779B97F4A7C15ull;
    return k ^ (k >> 29);
}

// returns the slot of `key`, inserting it if absent
__device__ __forceinline__ uint64_t hash_insert_slot(const HashView& h, uint64_t key) {
    uint64_t s = hash_mix(key) & h.mask;
    for (;;) {
        unsigned long long prev = atomicCAS(&h.keys[s], SEG3D_HASH_EMPTY, (unsigned long long)key);
        if (prev == SEG3D_HASH_EMPTY || prev == key) return s;
        s = (s + 1) & h.mask;
    }
}

__device__ __forceinline__ int32_t hash_lookup(const HashView& h, uint64_t key) {
    uint64_t s = hash_mix(key) & h.mask;
    for (;;) {
        unsigned long long cur = h.keys[s];
        if (cur == key) return h.vals[s];
        if (cur == SEG3D_HASH_EMPTY) return -1;
        s = (s + 1) & h.mask;
    }
}

// ---------------------------------------------------------------- scans (scan.hip)
// exclusive prefix sums over device arrays; tmp must hold scan_tmp_count(n) elements of T.
static inline size_t scan_tmp_count(int64_t n) { return (size_t)ceil_div64(n > 0 ? n : 1, 2048) + 1; }
int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total /*device, may be NULL*/,
                       uint32_t* tmp, hipStream_t st);
int scan_exclusive_u32x4(const uint4* in, uint4* out, int64_t n, uint4* total, uint4* tmp, hipStream_t st);

/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the integer / index
 * algorithms on the OpenSeg3D sparse-voxel hot path.  Never imported by the
 * product package (openseg3d_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Each function cites the reference lines (relative to /root/reference) whose
 * behaviour it restates.  Plain C99, single thread, no dependencies.
 *
 * Build: see oracle/Makefile  ->  oracle/_build/libseg3d_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small open-addressing map: int64 key -> int32 value (host side)     */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t *keys;
    int32_t *vals;
    uint64_t mask;
} omap;

static int omap_init(omap *m, int64_t n_items) {
    uint64_t cap = 16;
    while (cap < (uint64_t)(2 * n_items + 1)) cap <<= 1;
    m->keys = (int64_t *)malloc(cap * sizeof(int64_t));
    m->vals = (int32_t *)malloc(cap * sizeof(int32_t));
    if (!m->keys || !m->vals) return -1;
    for (uint64_t i = 0; i < cap; ++i) m->keys[i] = -1;
    m->mask = cap - 1;
    return 0;
}
static void omap_free(omap *m) {
    free(m->keys);
    free(m->vals);
}
static inline uint64_t omap_hash(int64_t k) {
    uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull;
    return x ^ (x >> 29);
}
/* returns pointer to the value slot; *found says whether key was present */
static inline int32_t *omap_slot(omap *m, int64_t k, int *found) {
    uint64_t s = omap_hash(k) & m->mask;
    for (;;) {
        if (m->keys[s] == k) { *found = 1; return &m->vals[s]; }
        if (m->keys[s] == -1) { *found = 0; m->keys[s] = k; return &m->vals[s]; }
        s = (s + 1) & m->mask;
    }
}
static inline int32_t omap_get(const omap *m, int64_t k) {
    uint64_t s = omap_hash(k) & m->mask;
    for (;;) {
        if (m->keys[s] == k) return m->vals[s];
        if (m->keys[s] == -1) return -1;
        s = (s + 1) & m->mask;
    }
}

/* ------------------------------------------------------------------ */
/* a1/a2: hard voxelisation, first-seen voxel order                    */
/* seg3d/core/voxel/voxel_generator.py:55-95 (points_to_voxel) and     */
/* :98-153 (_points_to_voxel_reverse_kernel).                          */
/*                                                                     */
/* The reference keeps a dense coor_to_voxelidx[z,y,x] int32 grid      */
/* (531 MB for 1440x1440x64); the lookup is replaced by a map keyed on */
/* the linear cell index -- same first-seen semantics, bounded memory. */
/* Arithmetic is done in the dtype of `points` exactly as written at   */
/* :139: floor((p - lo) / vs) with a true divide, compared against     */
/* grid_size (round((hi-lo)/vs), :129-132) as a float.                 */
/* ------------------------------------------------------------------ */
#define DEFINE_VOXELIZE(NAME, T, FLOOR, RINT)                                         \
    int64_t NAME(const T *points, int64_t n, int64_t stride, const T *voxel_size,     \
                 const T *coors_range, int32_t *coors_zyx /* [n,3] */,                \
                 int32_t *point_voxel_ids /* [n] */) {                                \
        int32_t grid[3];                                                              \
        for (int j = 0; j < 3; ++j) {                                                 \
            T g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];              \
            grid[j] = (int32_t)RINT(g); /* np.round = round-half-even */              \
        }                                                                             \
        omap m;                                                                       \
        if (omap_init(&m, n) != 0) return -1;                                         \
        int64_t voxel_num = 0;                                                        \
        for (int64_t i = 0; i < n; ++i) {                                             \
            int32_t coor[3];                                                          \
            int failed = 0;                                                           \
            point_voxel_ids[i] = -1;                                                  \
            for (int j = 0; j < 3; ++j) {                                             \
                T c = FLOOR((points[i * stride + j] - coors_range[j]) / voxel_size[j]); \
                if (c < 0 || c >= (T)grid[j]) { failed = 1; break; }                  \
                coor[2 - j] = (int32_t)c;                                             \
            }                                                                         \
            if (failed) continue;                                                     \
            int64_t key = ((int64_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2]; \
            int found;                                                                \
            int32_t *slot = omap_slot(&m, key, &found);                               \
            if (!found) {                                                             \
                *slot = (int32_t)voxel_num;                                           \
                coors_zyx[voxel_num * 3 + 0] = coor[0];                               \
                coors_zyx[voxel_num * 3 + 1] = coor[1];                               \
                coors_zyx[voxel_num * 3 + 2] = coor[2];                               \
                voxel_num++;                                                          \
            }                                                                         \
            point_voxel_ids[i] = *slot;                                               \
        }                                                                             \
        omap_free(&m);                                                                \
        return voxel_num;                                                             \
    }

DEFINE_VOXELIZE(oracle_voxelize_f32, float, floorf, rintf)
DEFINE_VOXELIZE(oracle_voxelize_f64, double, floor, rint)

/* ------------------------------------------------------------------ */
/* a14: rank of each element inside its group.                          */
/* seg3d/ops/ingroup_inds/src/ingroup_inds_cuda.cu:12-25 hands out      */
/* ranks in atomic arrival order (non-deterministic); the canonical     */
/* order used by this build is arrival in index order (stable rank).    */
/* ------------------------------------------------------------------ */
int oracle_ingroup_rank(const int64_t *group, int64_t n, int64_t *rank_out) {
    omap m;
    if (omap_init(&m, n) != 0) return -1;
    for (int64_t i = 0; i < n; ++i) {
        int found;
        int32_t *slot = omap_slot(&m, group[i], &found);
        if (!found) *slot = 0;
        rank_out[i] = (*slot)++;
    }
    omap_free(&m);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a9: submanifold 3x3x3 neighbour table ("rulebook").                  */
/* Call-site semantics of spconv.SubMConv3d(k=3, padding=1) as used at  */
/* seg3d/utils/spconv_utils.py:15-17 and pointtransformer.py:26-34:     */
/* output sites = input sites (same order), out[i] = sum_k W_k in[j]    */
/* where j is the active site at coords(i) + (kz-1, ky-1, kx-1).        */
/* nbr is [27][m] (offset-major), -1 = no active neighbour;             */
/* k = (kz*3 + ky)*3 + kx.  coords rows are [b, z, y, x] int32.         */
/* ------------------------------------------------------------------ */
static inline int64_t lin_key(int64_t b, int64_t z, int64_t y, int64_t x, const int32_t *shape) {
    return ((b * shape[0] + z) * shape[1] + y) * shape[2] + x;
}

int oracle_rulebook_subm(const int32_t *coords, int64_t m, const int32_t *shape_zyx,
                         int32_t *nbr /* [27][m] */) {
    omap map;
    if (omap_init(&map, m) != 0) return -1;
    for (int64_t i = 0; i < m; ++i) {
        int found;
        const int32_t *c = coords + 4 * i;
        *omap_slot(&map, lin_key(c[0], c[1], c[2], c[3], shape_zyx), &found) = (int32_t)i;
    }
    for (int k = 0; k < 27; ++k) {
        int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        for (int64_t i = 0; i < m; ++i) {
            const int32_t *c = coords + 4 * i;
            int z = c[1] + dz, y = c[2] + dy, x = c[3] + dx;
            int32_t j = -1;
            if (z >= 0 && z < shape_zyx[0] && y >= 0 && y < shape_zyx[1] && x >= 0 && x < shape_zyx[2])
                j = omap_get(&map, lin_key(c[0], z, y, x, shape_zyx));
            nbr[(int64_t)k * m + i] = j;
        }
    }
    omap_free(&map);
    return 0;
}

/* ------------------------------------------------------------------ */
/* a10: strided conv (k=3, s=2, p=1) output sites.                      */
/* spconv.SparseConv3d call site: spconv_utils.py:18-20,                */
/* pointtransformer.py:159-166.  out shape = floor((D+2-3)/2)+1;        */
/* output site o is active iff some active input sits at 2*o + k - 1.   */
/* Canonical (build-defined) order: ascending linear key (b,z,y,x).     */
/* Returns the number of output sites; out_coords must hold 8*m rows.   */
/* ------------------------------------------------------------------ */
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

int64_t oracle_downsample_coords(const int32_t *coords, int64_t m, const int32_t *shape_in,
                                 int32_t *out_coords /* [<=8m,4] */, int32_t *shape_out) {
    for (int j = 0; j < 3; ++j) shape_out[j] = (shape_in[j] + 2 - 3) / 2 + 1;
    int64_t *keys = (int64_t *)malloc((size_t)(8 * m + 1) * sizeof(int64_t));
    if (!keys) return -1;
    int64_t nk = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int32_t *c = coords + 4 * i;
        for (int k = 0; k < 27; ++k) {
            int kk[3] = {k / 9, (k / 3) % 3, k % 3};
            int o[3], ok = 1;
            for (int j = 0; j < 3; ++j) {
                int t = c[1 + j] + 1 - kk[j];
                if (t < 0 || (t & 1)) { ok = 0; break; }
                o[j] = t >> 1;
                if (o[j] >= shape_out[j]) { ok = 0; break; }
            }
            if (ok) keys[nk++] = lin_key(c[0], o[0], o[1], o[2], shape_out);
        }
    }
    qsort(keys, (size_t)nk, sizeof(int64_t), cmp_i64);
    int64_t mo = 0;
    for (int64_t i = 0; i < nk; ++i) {
        if (i && keys[i] == keys[i - 1]) continue;
        int64_t key = keys[i];
        int32_t *o = out_coords + 4 * mo++;
        o[3] = (int32_t)(key % shape_out[2]); key /= shape_out[2];
        o[2] = (int32_t)(key % shape_out[1]); key /= shape_out[1];
        o[1] = (int32_t)(key % shape_out[0]); key /= shape_out[0];
        o[0] = (int32_t)key;
    }
    free(keys);
    return mo;
}

/* ------------------------------------------------------------------ */
/* a10/a11: strided rulebook, both directions.                          */
/* nbr_fwd [27][m_out]: input row feeding output o through offset k     */
/*          (input site = 2*o + k - 1), -1 if inactive.                 */
/* nbr_inv [27][m_in] : output row o with 2*o + k - 1 == input site i   */
/*          -- the same (in, out, k) pairs with in/out swapped, which   */
/*          is what SparseInverseConv3d reuses under the same           */
/*          indice_key (pointtransformer.py:79-81).                     */
/* ------------------------------------------------------------------ */
int oracle_rulebook_strided(const int32_t *coords_in, int64_t m_in, const int32_t *shape_in,
                            const int32_t *coords_out, int64_t m_out, const int32_t *shape_out,
                            int32_t *nbr_fwd, int32_t *nbr_inv) {
    omap map;
    if (omap_init(&map, m_in) != 0) return -1;
    for (int64_t i = 0; i < m_in; ++i) {
        int found;
        const int32_t *c = coords_in + 4 * i;
        *omap_slot(&map, lin_key(c[0], c[1], c[2], c[3], shape_in), &found) = (int32_t)i;
    }
    for (int64_t i = 0; i < 27 * m_in; ++i) nbr_inv[i] = -1;
    for (int k = 0; k < 27; ++k) {
        int kk[3] = {k / 9, (k / 3) % 3, k % 3};
        for (int64_t o = 0; o < m_out; ++o) {
            const int32_t *c = coords_out + 4 * o;
            int p[3], ok = 1;
            for (int j = 0; j < 3; ++j) {
                p[j] = 2 * c[1 + j] + kk[j] - 1;
                if (p[j] < 0 || p[j] >= shape_in[j]) { ok = 0; break; }
            }
            int32_t i = ok ? omap_get(&map, lin_key(c[0], p[0], p[1], p[2], shape_in)) : -1;
            nbr_fwd[(int64_t)k * m_out + o] = i;
            if (i >= 0) nbr_inv[(int64_t)k * m_in + i] = (int32_t)o;
        }
    }
    omap_free(&map);
    return 0;
}

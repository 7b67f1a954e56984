/* CPU oracle, integer/byte part (SURVEY.md §8 rows H3 and the index half of H5).
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product library.
 * PARITY UNPINNED: the reference checkout has no source, tests or vectors for these
 * (src/latent_nerf is absent, scripts/train_latent_nerf.py:3-4); this restates the published
 * Morton interleave, the occupancy bit packing and the Instant-NGP spatial hash
 * (primes 1, 2654435761, 805459861) so that the HIP kernels and the PyTorch oracle
 * (oracle/nerf_oracle.py) can both be checked bit-for-bit against a third, scalar statement.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/libbits_oracle.so oracle/bits.c   (done by __graft_entry__.build()).
 */
#include <stdint.h>
#include <stddef.h>

static uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

static uint32_t compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xC30C30C3u;
    x = (x | (x >> 4)) & 0x0F00F00Fu;
    x = (x | (x >> 8)) & 0xFF0000FFu;
    x = (x | (x >> 16)) & 0x0000FFFFu;
    return x;
}

/* coords int32 [n,3] -> indices uint32 [n] ; x occupies bit 0 */
void oracle_morton3d(const int32_t *coords, uint32_t *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t x = (uint32_t)coords[3 * i], y = (uint32_t)coords[3 * i + 1], z = (uint32_t)coords[3 * i + 2];
        out[i] = expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
    }
}

void oracle_morton3d_invert(const uint32_t *idx, int32_t *coords, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        coords[3 * i] = (int32_t)compact_bits(idx[i]);
        coords[3 * i + 1] = (int32_t)compact_bits(idx[i] >> 1);
        coords[3 * i + 2] = (int32_t)compact_bits(idx[i] >> 2);
    }
}

/* grid float [n] (n % 8 == 0) -> bits uint8 [n/8]; bit k of byte b = grid[8b+k] > thresh */
void oracle_packbits(const float *grid, float thresh, uint8_t *bits, size_t n) {
    for (size_t b = 0; b < n / 8; ++b) {
        uint8_t v = 0;
        for (int k = 0; k < 8; ++k)
            if (grid[8 * b + k] > thresh) v |= (uint8_t)(1u << k);
        bits[b] = v;
    }
}

/* grid vertex (x,y,z) of a level with resolution `res` and `hashmap_size` rows -> row in level */
uint32_t oracle_grid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t hashmap_size) {
    uint32_t p[3] = {x, y, z};
    uint64_t stride = 1;
    uint32_t index = 0;
    for (int d = 0; d < 3 && stride <= hashmap_size; ++d) {
        index += p[d] * (uint32_t)stride;
        stride *= (uint64_t)(res + 1);
    }
    if (stride > hashmap_size)
        index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    return index % hashmap_size;
}

void oracle_grid_indices(const int32_t *verts, uint32_t *out, size_t n, uint32_t res, uint32_t hashmap_size) {
    for (size_t i = 0; i < n; ++i)
        out[i] = oracle_grid_index((uint32_t)verts[3 * i], (uint32_t)verts[3 * i + 1], (uint32_t)verts[3 * i + 2],
                                   res, hashmap_size);
}

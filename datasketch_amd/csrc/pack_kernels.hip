// pack_kernels.hip -- HBM-bound re-packing of signature matrices for downstream consumers:
// b-bit packing (bBitMinHash), MinHashLSH band keys (byte-swapped copy) and the LeanMinHash
// wire format.  One wave walks whole rows; lanes read consecutive uint64 values (coalesced).
#include "band_digest.h"
#include "mhx_internal.h"

namespace mhx {
namespace {

constexpr int kWave = 64;

// ---- b-bit packing --------------------------------------------------------------------------
// ref: datasketch/b_bit_minhash.py:37-38 (mask to b bits), :82-97 (value j of a block of
// n = 64/slot values sits at bit (n-1-j)*slot).  Lane l of a wave handles value kk0+l of a row;
// the `per` lanes of one block OR their shifted contributions together with a butterfly of
// DPP/shuffle steps and the first lane of the group stores the block.
template <typename SigT>
__global__ __launch_bounds__(256) void bbit_pack_kernel(const SigT *__restrict__ sig, int64_t n,
                                                        int32_t k, int32_t b, int32_t slot,
                                                        int32_t nb, uint64_t *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int per = 64 / slot;                    // values per block
    const int j = lane & (per - 1);               // position inside the block
    const int shift = (per - 1 - j) * slot;
    const uint64_t mask = b >= 32 ? 0xFFFFFFFFull : ((1ull << b) - 1ull);
    const int padded = nb * per;                  // row length rounded up to whole blocks
    const int blocks_per_iter = kWave / per;      // blocks a wave finishes per 64 values
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n;
         row += (int64_t)gridDim.x * waves_per_block) {
        const SigT *src = sig + row * k;
        uint64_t *dst = out + row * nb;
        for (int kk0 = 0; kk0 < padded; kk0 += kWave) {
            const int kk = kk0 + lane;
            if (slot == 1) {
                // one bit per value: the wave's ballot IS the block, lane 0 holding the top bit
                // (3.0 -> 4.9 TB/s; the shuffle butterfly below is what bounds the other widths)
                const uint64_t bits = __brevll(__ballot(kk < k && (src[kk] & 1u) != 0));
                if (lane == 0) dst[kk0 >> 6] = bits;
                continue;
            }
            uint64_t v = 0;
            if (kk < k) v = ((uint64_t)(uint32_t)((uint64_t)src[kk] & mask)) << shift;
            // OR-reduce across the `per` lanes of the block
            for (int d = 1; d < per; d <<= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)v, d);
                const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), d);
                v |= ((uint64_t)hi << 32) | lo;
            }
            const int blk = kk0 / per + lane / per;
            if (j == 0 && blk < nb && lane / per < blocks_per_iter) dst[blk] = v;
        }
    }
}

// b = 1 with 16-byte loads.  The kernel above reads one value per lane per load instruction; for uint32 signatures (the
// compact / all-gathered form) that is 256 B per instruction and the kernel ran at 3.0 TB/s.  Here a lane loads 16 B =
// V consecutive values (4 uint32 or 2 uint64), turns their low bits into its V bits of the block -- value j of a block
// sits at bit 63 - j, so lane i of the 64/V lanes of a block owns bits 63 - V*i .. 64 - V*(i + 1) -- and the lanes
// of a block OR their words together with DPP rotations inside the 16-lane row (plus one cross-row step for uint64).
// One load and one 8-byte store per lane group per 64 values; needs k % (64 * V) == 0 and 16-byte aligned rows.
__device__ __forceinline__ uint32_t or_row16(uint32_t x) {  // OR over the 16 lanes of a DPP row, in every lane
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xF, 0xF, false);  // row_ror:1
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xF, 0xF, false);  // row_ror:2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xF, 0xF, false);  // row_ror:4
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);  // row_ror:8
    return x;
}

template <typename SigT>
__global__ __launch_bounds__(256) void bbit1_wide_kernel(const SigT *__restrict__ sig, int64_t n, int32_t k,
                                                         uint64_t *__restrict__ out) {
    constexpr int V = 16 / (int)sizeof(SigT);  // values per lane
    constexpr int L = kWave / V;               // lanes per block of 64 values
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int i = lane & (L - 1);              // position of the lane inside its block
    const int nb = k / 64;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n; row += (int64_t)gridDim.x * waves_per_block) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sig + row * k);
        uint64_t *dst = out + row * nb;
        for (int c0 = 0; c0 < k / V; c0 += kWave) {  // 64 * V values = V blocks per step
            const uint4 v = src[c0 + lane];
            uint32_t bits;  // the lane's V bits, first value highest
            if (V == 4)
                bits = ((v.x & 1u) << 3) | ((v.y & 1u) << 2) | ((v.z & 1u) << 1) | (v.w & 1u);
            else
                bits = ((v.x & 1u) << 1) | (v.z & 1u);  // low words of the two uint64
            const int sh = 64 - V * (i + 1);            // bit position of the lane's last value
            uint32_t hi = sh >= 32 ? bits << (sh - 32) : 0u;
            uint32_t lo = sh >= 32 ? 0u : bits << sh;
            hi = or_row16(hi);
            lo = or_row16(lo);
            if (L == 32) {  // a block spans two rows of 16 lanes
                hi |= __shfl_xor(hi, 16);
                lo |= __shfl_xor(lo, 16);
            }
            if (i == 0) dst[(c0 * V) / 64 + lane / L] = ((uint64_t)hi << 32) | lo;
        }
    }
}

// The same idea for the wider slots (2, 4, 8, 16 and, with uint64 input, 32 bits per value): a lane's V values go to
// their slots of a 64-bit word, the per/V lanes of a block (16, 8, 4, 2 or 1) OR their words with DPP butterflies
// (mirror inside the row / half row, then the two quad permutations), the first lane of the group stores.
template <int LB>
__device__ __forceinline__ uint32_t or_group(uint32_t x) {  // OR over aligned groups of LB lanes (LB <= 16), in every lane
    if constexpr (LB >= 16) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);  // row_mirror
    if constexpr (LB >= 8) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);   // row_half_mirror
    if constexpr (LB >= 4) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    if constexpr (LB >= 2) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    return x;
}

template <typename SigT, int SLOT>
__global__ __launch_bounds__(256) void bbit_wide_kernel(const SigT *__restrict__ sig, int64_t n, int32_t k, int32_t b,
                                                        uint64_t *__restrict__ out) {
    constexpr int V = 16 / (int)sizeof(SigT);  // values per lane
    constexpr int PER = 64 / SLOT;             // values per block
    constexpr int LB = PER / V;                // lanes per block (>= 1: the launcher sees to it)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int i = lane & (LB - 1);
    const uint32_t mask = b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u);
    const int nb = k / PER;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n; row += (int64_t)gridDim.x * waves_per_block) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sig + row * k);
        uint64_t *dst = out + row * nb;
        for (int c0 = 0; c0 < k / V; c0 += kWave) {
            const uint4 v = src[c0 + lane];
            uint64_t word;
            if (V == 4) {
                const int top = 64 - SLOT * (4 * i + 1);  // shift of the lane's first value
                word = ((uint64_t)(v.x & mask) << top) | ((uint64_t)(v.y & mask) << (top - SLOT)) |
                       ((uint64_t)(v.z & mask) << (top - 2 * SLOT)) | ((uint64_t)(v.w & mask) << (top - 3 * SLOT));
            } else {
                const int top = 64 - SLOT * (2 * i + 1);
                word = ((uint64_t)(v.x & mask) << top) | ((uint64_t)(v.z & mask) << (top - SLOT));
            }
            const uint32_t hi = or_group<LB>((uint32_t)(word >> 32)), lo = or_group<LB>((uint32_t)word);
            if (i == 0) dst[(c0 * V) / PER + lane / LB] = ((uint64_t)hi << 32) | lo;
        }
    }
}

// ---- band keys ------------------------------------------------------------------------------
// ref: datasketch/lsh.py:537-538 (_byteswap) over hashranges (:199): out[i, c] = bswap64(sig[i, c])
__global__ __launch_bounds__(256) void band_keys_kernel(const uint64_t *__restrict__ sig, int64_t n,
                                                        int32_t k, int32_t w,
                                                        uint64_t *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n;
         row += (int64_t)gridDim.x * waves_per_block) {
        const uint64_t *src = sig + row * k;
        uint64_t *dst = out + row * w;
        for (int c = lane; c < w; c += kWave) dst[c] = __builtin_bswap64(src[c]);
    }
}

// flat variant when the whole matrix is converted (w == k): 16 B per lane, four loads in flight, non-temporal both ways,
// 32 workgroups per CU -- the best of the recipes in tools/ubench_stream.hip (profiles/r02_ubench_stream_recipes.jsonl:
// 5.1 TB/s read + write against 4.65 for the plain one-load loop at 8 workgroups per CU)
__device__ __forceinline__ ulonglong2 load_nt(const ulonglong2 *p) {
    ulonglong2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
}
__device__ __forceinline__ void store_nt(ulonglong2 *p, ulonglong2 v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}

__global__ __launch_bounds__(256) void bswap_flat_kernel(const ulonglong2 *__restrict__ in, int64_t n2,
                                                         ulonglong2 *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 4 * stride) {
        ulonglong2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < n2) v[u] = load_nt(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride >= n2) break;
            v[u].x = __builtin_bswap64(v[u].x);
            v[u].y = __builtin_bswap64(v[u].y);
            store_nt(out + i + u * stride, v[u]);
        }
    }
}

// ---- LeanMinHash wire format ----------------------------------------------------------------
// ref: datasketch/lean_minhash.py:174-175: struct "<byteorder>qi{K}I" = seed(int64) K(int32) K x uint32 -- no padding in any of
// the byte orders the reference accepts ('@' and '=' are this host's little-endian, '<'; '>' and '!' big-endian).
// A record is (3 + K) 32-bit words; records are only 4-byte aligned, so loads and stores are dword-wide.
template <typename SigT, bool BIG>
__global__ __launch_bounds__(256) void lean_serialize_kernel(const SigT *__restrict__ sig, int64_t n,
                                                             int32_t k, int64_t seed,
                                                             uint32_t *__restrict__ out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int rec = 3 + k;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n;
         row += (int64_t)gridDim.x * waves_per_block) {
        const SigT *src = sig + row * k;
        uint32_t *dst = out + row * rec;
        if (lane == 0) {
            const uint32_t lo = (uint32_t)seed, hi = (uint32_t)((uint64_t)seed >> 32);
            dst[0] = BIG ? __builtin_bswap32(hi) : lo;
            dst[1] = BIG ? __builtin_bswap32(lo) : hi;
            dst[2] = BIG ? __builtin_bswap32((uint32_t)k) : (uint32_t)k;
        }
        for (int c = lane; c < k; c += kWave) dst[3 + c] = BIG ? __builtin_bswap32((uint32_t)src[c]) : (uint32_t)src[c];
    }
}

// LeanMinHash.deserialize of n records (ref: datasketch/lean_minhash.py:177-214): hashvalues widened to SigT, the record's seed to
// seeds[row]; a record whose length field is not k is counted in *bad (its k values are decoded all the same: the caller decides).
template <typename SigT, bool BIG>
__global__ __launch_bounds__(256) void lean_deserialize_kernel(const uint32_t *__restrict__ records, int64_t n, int32_t k,
                                                               SigT *__restrict__ sig, int64_t *__restrict__ seeds,
                                                               unsigned int *__restrict__ bad) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int rec = 3 + k;
    for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave; row < n;
         row += (int64_t)gridDim.x * waves_per_block) {
        const uint32_t *src = records + row * rec;
        SigT *dst = sig + row * k;
        if (lane == 0) {
            const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
            const uint64_t seed = BIG ? ((uint64_t)__builtin_bswap32(w0) << 32) | __builtin_bswap32(w1) : ((uint64_t)w1 << 32) | w0;
            if (seeds) seeds[row] = (int64_t)seed;
            if (bad && (BIG ? __builtin_bswap32(w2) : w2) != (uint32_t)k) atomicAdd(bad, 1u);
        }
        for (int c = lane; c < k; c += kWave) dst[c] = (SigT)(BIG ? __builtin_bswap32(src[3 + c]) : src[3 + c]);
    }
}

// bBitMinHash.__setstate__ of every row (ref: datasketch/b_bit_minhash.py:103-125): value j of block i is
// (block >> (n - 1 - j) * slot) & (2^slot - 1), n = 64 / slot; out[row, i * n + j] as uint32.  A thread produces four
// consecutive values (one 16-byte store when the rows allow it); the blocks of a row are a few cache lines read by all.
template <bool WIDE>
__global__ __launch_bounds__(256) void bbit_unpack_kernel(const uint64_t *__restrict__ blocks, int64_t n, int32_t k, int32_t slot,
                                                          int32_t nb, uint32_t *__restrict__ out) {
    const int per_log2 = 6 - (31 - __builtin_clz((unsigned)slot));  // values per block = 64 / slot, a power of two
    const int per = 1 << per_log2;
    const uint64_t mask = slot >= 64 ? ~0ull : ((1ull << slot) - 1ull);
    const int groups = (k + 3) >> 2;  // groups of four values per row
    const int64_t total = n * (int64_t)groups;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / groups;
        const int kk = (int)(idx - row * groups) << 2;
        const uint64_t *src = blocks + row * nb;
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int at = kk + u < k ? kk + u : k - 1;
            const uint64_t blk = src[at >> per_log2];
            v[u] = (uint32_t)((blk >> ((per - 1 - (at & (per - 1))) * slot)) & mask);
        }
        uint32_t *dst = out + row * k + kk;
        if (WIDE) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (kk + u < k) dst[u] = v[u];
        }
    }
}

// ---- band digests -----------------------------------------------------------------------------
// out[i, j] = FNV-1a-64 of the band key of band j of row i, i.e. of exactly the bytes the reference
// uses as that band's dictionary key (ref: datasketch/lsh.py:199,344,537-538: the r hashvalues of the
// band, each as 8 big-endian bytes).  It is what MinHashLSH(hashfunc=fnv1a_64) would store
// (ref: lsh.py:540-543), in a form a device-side sort can group by.  One lane per (row, band).
// (shift >= 0: bands is that power of two -- the 64-bit division of idx by a run-time divisor was 60 of the kernel's ~290
// VALU instructions per element; a run-time shift costs what a compile-time one does)
template <typename SigT>
__global__ __launch_bounds__(256) void band_digest_kernel(const SigT *__restrict__ sig, int64_t n, int32_t k,
                                                          int32_t bands, int32_t r, int shift, int band_major, uint64_t *__restrict__ out) {
    const int64_t total = n * (int64_t)bands;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = shift >= 0 ? idx >> shift : idx / bands;
        const int band = (int)(idx - row * bands);
        out[band_major ? (int64_t)band * n + row : idx] = band_digest_of<SigT>(sig, row, band, k, r);
    }
}

// Band-major output ([bands, n]) through LDS.  A lane owns one (row, band), so written straight out a wave's 64 digests go to
// `bands` different streams, 16 bytes each (measured: band_digest_kernel 0.31 -> 0.51 ms, the fused kernel 0.34 -> 0.49).
// Here a workgroup takes kTileIters x 256 consecutive (row, band) pairs -- 64 whole rows at 32 bands --, parks the digests in
// LDS as [band][row of the tile] and writes every band's run of the tile (512 B at 32 bands) with consecutive lanes.
constexpr int kTileIters = 8;
template <int BANDS_LOG2_MAX = 6>
struct BandMajorTile {
    // digests of one tile: [bands][rows + 1] (one uint64 of padding per band: lanes of a wave differ in the band first)
    __device__ static __forceinline__ int rows(int shift) { return (kTileIters * 256) >> shift; }
    __device__ static __forceinline__ void put(uint64_t *lds, int shift, int64_t local, uint64_t h) {  // local = idx - tile base
        const int band = (int)local & ((1 << shift) - 1), row = (int)(local >> shift);
        lds[band * (rows(shift) + 1) + row] = h;
    }
    // after a barrier: the tile's digests to out[band * n + row0 + row], consecutive lanes along a band's run
    __device__ static __forceinline__ void flush(const uint64_t *lds, int shift, int64_t row0, int64_t n, uint64_t *__restrict__ out) {
        const int tr = rows(shift), count = kTileIters * 256;
        for (int e = threadIdx.x; e < count; e += 256) {
            const int band = e / tr, row = e - band * tr;
            if (row0 + row < n) out[(int64_t)band * n + row0 + row] = lds[band * (tr + 1) + row];
        }
    }
};

template <typename SigT>
__global__ __launch_bounds__(256) void band_digest_bm_kernel(const SigT *__restrict__ sig, int64_t n, int32_t k, int32_t r, int shift,
                                                             uint64_t *__restrict__ out) {
    __shared__ uint64_t tile[kTileIters * 256 + 64];
    const int bands = 1 << shift;
    const int64_t total = n << shift;
    constexpr int64_t kTile = kTileIters * 256;
    for (int64_t base = (int64_t)blockIdx.x * kTile; base < total; base += (int64_t)gridDim.x * kTile) {
#pragma unroll 2
        for (int it = 0; it < kTileIters; ++it) {
            const int64_t idx = base + it * 256 + threadIdx.x;
            if (idx < total) {
                const int64_t row = idx >> shift;
                BandMajorTile<>::put(tile, shift, idx - base, band_digest_of<SigT>(sig, row, (int)(idx & (bands - 1)), k, r));
            }
        }
        __syncthreads();
        BandMajorTile<>::flush(tile, shift, base >> shift, n, out);
        __syncthreads();
    }
}

// ---- b-bit packing AND band digests from ONE read of the matrix (config 5) ---------------------------------------
// ref: datasketch/b_bit_minhash.py:78-101 (blocks) and lsh.py:199,344,537-543 (band keys, hashed).  Config 5 wants both of
// a 10M x 256 matrix; as two kernels the 1 KB rows cross the HBM bus twice (2.9 GB of traffic for 1.64 GB of algorithmic
// bytes per 1.25M rows).  Here one lane owns one (row, band): it loads the band's R values once (R * sizeof(SigT)
// contiguous bytes, 16-byte loads), hashes them (FNV-1a-64 of the reference's key bytes, as band_digest_of) and turns the
// same registers into its part of the row's b-bit blocks:
//   * R >= 64 / slot: the lane's values fill R * slot / 64 whole blocks, stored directly;
//   * R <  64 / slot: G = 64 / (slot * R) neighbouring lanes (bands) share a block; each shifts its R * slot bits into
//     place and the group ORs the words with DPP steps (shuffles for G = 32, 64), the group's first lane stores.
// Shapes: bands a power of two (<= 64: a wave holds whole rows), R in {4, 8, 16}, bands * R == num_perm (every value of the
// row belongs to a band, so the blocks are complete), bands % G == 0, 16-byte aligned rows.  launch_bbit_digest_fused says
// whether a shape qualifies; the caller runs the two kernels otherwise.
template <int G>
__device__ __forceinline__ uint32_t or_lanes(uint32_t x) {  // OR over aligned groups of G lanes, in every lane
    if constexpr (G <= 16) {
        return or_group<G>(x);
    } else {
        x = or_group<16>(x);
        x |= (uint32_t)__shfl_xor((int)x, 16);
        if constexpr (G == 64) x |= (uint32_t)__shfl_xor((int)x, 32);
        return x;
    }
}

template <typename SigT, int SLOT, int R>
__global__ __launch_bounds__(256) void bbit_digest_fused_kernel(const SigT *__restrict__ sig, int64_t n, int32_t b, int band_shift, int band_major,
                                                                uint64_t *__restrict__ blocks, uint64_t *__restrict__ digests) {
    constexpr int PER = 64 / SLOT;               // values per block
    constexpr int G = PER > R ? PER / R : 1;     // lanes per block
    constexpr int Q = R >= PER ? R / PER : 1;    // blocks per lane
    constexpr bool kWide = sizeof(SigT) == 8;
    // bands * R == num_perm (the launcher sees to it), so with idx = row * bands + band:
    //   the band's values start at sig + idx * R, its digest goes to digests[idx], and -- a row holding bands / G (or
    //   bands * Q) blocks -- its block is blocks[idx / G] (or blocks[idx * Q + q]): no row / band arithmetic at all
    const int64_t total = n << band_shift;
    const uint32_t mask = b >= 32 ? 0xFFFFFFFFu : ((1u << b) - 1u);
    // band-major digests go through an LDS tile of kTileIters x 256 consecutive (row, band) pairs (BandMajorTile); row-major
    // ones straight out.  Either way a workgroup walks whole tiles, so that the two variants read the matrix in the same order.
    __shared__ uint64_t tile[kTileIters * 256 + 64];
    constexpr int64_t kTile = kTileIters * 256;
    for (int64_t tbase = (int64_t)blockIdx.x * kTile; tbase < total; tbase += (int64_t)gridDim.x * kTile) {
    for (int it = 0; it < kTileIters; ++it) {
        const int64_t idx = tbase + it * 256 + threadIdx.x;
        const bool live = idx < total;           // (total is a multiple of G: a group is live or not as a whole)
        uint32_t lo[R], hi[R];
        if (live) {
            const SigT *src = sig + idx * R;
            if constexpr (kWide) {
                const ulonglong2 *s2 = reinterpret_cast<const ulonglong2 *>(src);
#pragma unroll
                for (int c = 0; c < R / 2; ++c) {
                    const ulonglong2 v = s2[c];
                    lo[2 * c] = (uint32_t)v.x, hi[2 * c] = (uint32_t)(v.x >> 32);
                    lo[2 * c + 1] = (uint32_t)v.y, hi[2 * c + 1] = (uint32_t)(v.y >> 32);
                }
            } else {
                const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
#pragma unroll
                for (int c = 0; c < R / 4; ++c) {
                    const uint4 v = s4[c];
                    lo[4 * c] = v.x, lo[4 * c + 1] = v.y, lo[4 * c + 2] = v.z, lo[4 * c + 3] = v.w;
                }
#pragma unroll
                for (int i = 0; i < R; ++i) hi[i] = 0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) lo[i] = 0, hi[i] = 0;
        }
        // the digest of the band's key
        uint32_t h_hi = 0xcbf29ce4u, h_lo = 0x84222325u;
#pragma unroll
        for (int i = 0; i < R; ++i) fnv_absorb_value(h_hi, h_lo, kWide ? hi[i] : 0u, lo[i]);
        if (live) {
            const uint64_t h = ((uint64_t)h_hi << 32) | h_lo;
            if (band_major) BandMajorTile<>::put(tile, band_shift, idx - tbase, h);
            else digests[idx] = h;
        }
        // the band's part of the row's blocks (value j of a block sits at bit (PER - 1 - j) * SLOT)
        if constexpr (G > 1) {
            constexpr int kBits = R * SLOT;      // the lane's R values, first highest: < 64 bits
            const int g = (int)idx & (G - 1);    // (= band % G: bands is a multiple of G)
            const int sh = (G - 1 - g) * kBits;  // where the lane's bits sit in the block
            uint32_t whi, wlo;
            if constexpr (SLOT == 1) {
                // b = 1: a funnel shift per value (v_alignbit_b32: acc = {value, acc} >> 1 moves the value's low bit in at
                // the top), last value first -- the lane's R bits end up top-aligned, first value highest; one 64-bit
                // shift puts them at bit 63 - g * R of the block
                uint32_t acc = 0;
#pragma unroll
                for (int i = R - 1; i >= 0; --i) acc = __builtin_amdgcn_alignbit(lo[i], acc, 1);
                const uint64_t word = ((uint64_t)acc << 32) >> (g * R);
                whi = (uint32_t)(word >> 32), wlo = (uint32_t)word;
            } else if constexpr (kBits <= 32) {  // 32-bit arithmetic: the chunk lands in one half of the block
                uint32_t chunk = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) chunk |= (lo[i] & mask) << ((R - 1 - i) * SLOT);
                whi = sh >= 32 ? chunk << (sh - 32) : 0u;
                wlo = sh >= 32 ? 0u : chunk << sh;
            } else {
                uint64_t chunk = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) chunk |= (uint64_t)(lo[i] & mask) << ((R - 1 - i) * SLOT);
                const uint64_t word = chunk << sh;
                whi = (uint32_t)(word >> 32), wlo = (uint32_t)word;
            }
            whi = or_lanes<G>(whi), wlo = or_lanes<G>(wlo);
            if (live && g == 0) blocks[idx / G] = ((uint64_t)whi << 32) | wlo;
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                uint64_t word = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) word |= (uint64_t)(lo[q * PER + i] & mask) << ((PER - 1 - i) * SLOT);
                if (live) blocks[idx * Q + q] = word;
            }
        }
    }
        if (band_major) {  // (kernel argument: workgroup-uniform)
            __syncthreads();
            BandMajorTile<>::flush(tile, band_shift, tbase >> band_shift, n, digests);
            __syncthreads();
        }
    }
}

// ---- batched Jaccard estimate -----------------------------------------------------------------
// counts[p] = number of positions where signature rows pairs[p][0] (of A) and pairs[p][1] (of B) agree;
// MinHash.jaccard is counts / K (ref: datasketch/minhash.py:299-324).  One wave per pair, lanes
// over the K positions (coalesced row reads), ballot + popcount.
template <typename SigT>
__global__ __launch_bounds__(256) void jaccard_pairs_kernel(const SigT *__restrict__ a, const SigT *__restrict__ b,
                                                            int32_t k, const int64_t *__restrict__ pairs, int64_t m,
                                                            int32_t *__restrict__ counts) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    for (int64_t p = (int64_t)blockIdx.x * waves_per_block + wave; p < m; p += (int64_t)gridDim.x * waves_per_block) {
        const SigT *x = a + pairs[2 * p] * k;
        const SigT *y = b + pairs[2 * p + 1] * k;
        int32_t cnt = 0;
        for (int c = lane; c < k + lane; c += kWave) {  // uniform trip count: every lane reaches the ballot
            const bool eq = c < k && x[c] == y[c];
            cnt += __popcll(__ballot(eq));
        }
        if (lane == 0) counts[p] = cnt;
    }
}

// ---- batched b-bit Jaccard numerators -----------------------------------------------------------
// counts[p] = number of positions whose b-bit values agree in packed rows pairs[p][0] (of A) and pairs[p][1] (of B):
// the `intersection` of bBitMinHash.jaccard (ref: datasketch/b_bit_minhash.py:53-72), taken on the packed blocks
// (ref :82-101) without unpacking: z = x ^ y, OR every slot's bits down into its lowest bit, popcount the slots
// that differ.  Slots behind num_perm are zero in both rows, so agreeing positions = num_perm - differing slots.
// LPP lanes share a pair (LPP = power of two >= blocks per row, at most 64): coalesced row reads, shuffle reduction.
__global__ __launch_bounds__(256) void bbit_jaccard_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                                                           int32_t nb, int32_t slot, int32_t k, int32_t lpp,
                                                           const int64_t *__restrict__ pairs, int64_t m,
                                                           int32_t *__restrict__ counts) {
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane & (lpp - 1);          // my position among the lanes of one pair
    const int pairs_per_wave = kWave / lpp;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t wave_stride = (int64_t)gridDim.x * (blockDim.x >> 6);
    // lowest bit of every slot
    const uint64_t low = slot == 1 ? ~0ull : slot == 2 ? 0x5555555555555555ull : slot == 4 ? 0x1111111111111111ull
                       : slot == 8 ? 0x0101010101010101ull : slot == 16 ? 0x0001000100010001ull : 0x0000000100000001ull;
    for (int64_t p0 = wave_id * pairs_per_wave; p0 < m; p0 += wave_stride * pairs_per_wave) {
        const int64_t p = p0 + lane / lpp;
        int32_t differ = 0;
        if (p < m) {
            const uint64_t *x = a + pairs[2 * p] * nb, *y = b + pairs[2 * p + 1] * nb;
            for (int c = sub; c < nb; c += lpp) {
                uint64_t z = x[c] ^ y[c];
                for (int sh = 1; sh < slot; sh <<= 1) z |= z >> sh;
                differ += __popcll(z & low);
            }
        }
        for (int d = 1; d < lpp; d <<= 1) differ += __shfl_xor(differ, d);
        if (p < m && sub == 0) counts[p] = k - differ;
    }
}

inline dim3 row_grid(mhx_ctx *ctx, int64_t n) {
    const int64_t want = (n + 3) / 4;
    return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 8)));
}

}  // namespace

int launch_bbit_pack(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b,
                     uint64_t *d_out) {
    const int slot = bbit_slot_size(b);
    const int per = 64 / slot;
    const int nb = (k + per - 1) / per;
    const int esize = sig_dtype == MHX_U32 ? 4 : 8;
    if (slot == 1 && k % (64 * (16 / esize)) == 0 && ((uintptr_t)d_sig & 15) == 0) {  // 16-byte loads
        if (sig_dtype == MHX_U32)
            hipLaunchKernelGGL(bbit1_wide_kernel<uint32_t>, row_grid(ctx, n), dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n, k, d_out);
        else
            hipLaunchKernelGGL(bbit1_wide_kernel<uint64_t>, row_grid(ctx, n), dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n, k, d_out);
    } else if (slot > 1 && 64 / slot >= 16 / esize && k % (64 * (16 / esize)) == 0 && ((uintptr_t)d_sig & 15) == 0) {
        const dim3 g = row_grid(ctx, n);
#define MHX_WIDE(T, S) hipLaunchKernelGGL((bbit_wide_kernel<T, S>), g, dim3(256), 0, ctx->stream, (const T *)d_sig, n, k, b, d_out)
        if (sig_dtype == MHX_U32) {
            if (slot == 2) MHX_WIDE(uint32_t, 2); else if (slot == 4) MHX_WIDE(uint32_t, 4);
            else if (slot == 8) MHX_WIDE(uint32_t, 8); else MHX_WIDE(uint32_t, 16);
        } else {
            if (slot == 2) MHX_WIDE(uint64_t, 2); else if (slot == 4) MHX_WIDE(uint64_t, 4);
            else if (slot == 8) MHX_WIDE(uint64_t, 8); else if (slot == 16) MHX_WIDE(uint64_t, 16); else MHX_WIDE(uint64_t, 32);
        }
#undef MHX_WIDE
    } else
    if (sig_dtype == MHX_U32)
        hipLaunchKernelGGL(bbit_pack_kernel<uint32_t>, row_grid(ctx, n), dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n,
                           k, b, slot, nb, d_out);
    else
        hipLaunchKernelGGL(bbit_pack_kernel<uint64_t>, row_grid(ctx, n), dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n,
                           k, b, slot, nb, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

// One pass for both (see bbit_digest_fused_kernel); *done = false when the shape does not qualify (nothing launched).
int launch_bbit_digest_fused(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b, int32_t bands,
                             int32_t r, uint64_t *d_blocks, uint64_t *d_digests, int layout, bool *done) {
    *done = false;
    const int slot = bbit_slot_size(b);
    const int per = 64 / slot;
    const int esize = sig_dtype == MHX_U32 ? 4 : 8;
    if (!(r == 4 || r == 8 || r == 16) || bands > 64 || (bands & (bands - 1)) != 0 || (int64_t)bands * r != k) return MHX_OK;
    const int g = per > r ? per / r : 1;
    if (bands % g != 0 || k % per != 0 || (r * esize) % 16 != 0) return MHX_OK;
    if ((((uintptr_t)d_sig) & 15) != 0 || (((int64_t)k * esize) & 15) != 0) return MHX_OK;
    int shift = 0;
    while ((1 << shift) < bands) ++shift;
    const int64_t want = (n * bands + kTileIters * 256 - 1) / (kTileIters * 256);
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 16)));
#define MHX_FUSED(T, S, RR) hipLaunchKernelGGL((bbit_digest_fused_kernel<T, S, RR>), grid, dim3(256), 0, ctx->stream, (const T *)d_sig, n, b, shift, layout, d_blocks, d_digests)
#define MHX_FUSED_R(T, S)                                                              \
    do {                                                                               \
        if (r == 4) MHX_FUSED(T, S, 4); else if (r == 8) MHX_FUSED(T, S, 8); else MHX_FUSED(T, S, 16); \
    } while (0)
#define MHX_FUSED_S(T)                                                                 \
    do {                                                                               \
        switch (slot) {                                                                \
            case 1: MHX_FUSED_R(T, 1); break;                                          \
            case 2: MHX_FUSED_R(T, 2); break;                                          \
            case 4: MHX_FUSED_R(T, 4); break;                                          \
            case 8: MHX_FUSED_R(T, 8); break;                                          \
            case 16: MHX_FUSED_R(T, 16); break;                                        \
            default: MHX_FUSED_R(T, 32); break;                                        \
        }                                                                              \
    } while (0)
    if (sig_dtype == MHX_U32)
        MHX_FUSED_S(uint32_t);
    else
        MHX_FUSED_S(uint64_t);
#undef MHX_FUSED_S
#undef MHX_FUSED_R
#undef MHX_FUSED
    MHX_HIP_CHECK(hipGetLastError());
    *done = true;
    return MHX_OK;
}

int launch_band_keys(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                     uint64_t *d_out) {
    const int w = bands * r;
    const int64_t total = n * (int64_t)k;
    const bool aligned = (((uintptr_t)d_sig | (uintptr_t)d_out) & 15) == 0;
    if (w == k && (total & 1) == 0 && aligned) {
        const int64_t n2 = total >> 1;
        const int64_t want = (n2 + 1023) / 1024;  // four 16-byte items per thread and trip
        dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 32)));
        hipLaunchKernelGGL(bswap_flat_kernel, grid, dim3(256), 0, ctx->stream, (const ulonglong2 *)d_sig, n2,
                           (ulonglong2 *)d_out);
    } else {
        hipLaunchKernelGGL(band_keys_kernel, row_grid(ctx, n), dim3(256), 0, ctx->stream, d_sig, n, k, w, d_out);
    }
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_band_digests(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands, int32_t r,
                        uint64_t *d_out, int layout) {
    const int64_t want = (n * bands + 255) / 256;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 16)));
    const int shift = (bands & (bands - 1)) == 0 ? __builtin_ctz((unsigned)bands) : -1;
    if (layout == MHX_BAND_MAJOR && shift >= 0 && bands <= 64) {  // through an LDS tile (any other band count: straight out)
        const int64_t tiles = (n * bands + kTileIters * 256 - 1) / (kTileIters * 256);
        dim3 tgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)ctx->num_cus * 16)));
        if (sig_dtype == MHX_U32)
            hipLaunchKernelGGL(band_digest_bm_kernel<uint32_t>, tgrid, dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n, k, r, shift, d_out);
        else
            hipLaunchKernelGGL(band_digest_bm_kernel<uint64_t>, tgrid, dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n, k, r, shift, d_out);
        MHX_HIP_CHECK(hipGetLastError());
        return MHX_OK;
    }
    if (sig_dtype == MHX_U32)
        hipLaunchKernelGGL(band_digest_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n, k, bands, r, shift, layout, d_out);
    else
        hipLaunchKernelGGL(band_digest_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n, k, bands, r, shift, layout, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_jaccard_pairs(mhx_ctx *ctx, const void *d_a, const void *d_b, int sig_dtype, int32_t k, const int64_t *d_pairs,
                         int64_t m, int32_t *d_counts) {
    if (sig_dtype == MHX_U32)
        hipLaunchKernelGGL(jaccard_pairs_kernel<uint32_t>, row_grid(ctx, m), dim3(256), 0, ctx->stream, (const uint32_t *)d_a,
                           (const uint32_t *)d_b, k, d_pairs, m, d_counts);
    else
        hipLaunchKernelGGL(jaccard_pairs_kernel<uint64_t>, row_grid(ctx, m), dim3(256), 0, ctx->stream, (const uint64_t *)d_a,
                           (const uint64_t *)d_b, k, d_pairs, m, d_counts);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_bbit_jaccard(mhx_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, int32_t k, int32_t b, const int64_t *d_pairs,
                        int64_t m, int32_t *d_counts) {
    const int slot = bbit_slot_size(b);
    const int per = 64 / slot;
    const int nb = (k + per - 1) / per;
    int lpp = 1;
    while (lpp < nb && lpp < kWave) lpp <<= 1;
    const int64_t waves = (m * lpp + kWave - 1) / kWave;
    const int64_t want = (waves + 3) / 4;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 8)));
    hipLaunchKernelGGL(bbit_jaccard_kernel, grid, dim3(256), 0, ctx->stream, d_a, d_b, nb, slot, k, lpp, d_pairs, m, d_counts);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_lean_serialize(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int64_t seed, int big_endian,
                          uint8_t *d_out) {
    const dim3 g = row_grid(ctx, n);
    uint32_t *out = (uint32_t *)d_out;
    if (sig_dtype == MHX_U32) {
        if (big_endian) hipLaunchKernelGGL((lean_serialize_kernel<uint32_t, true>), g, dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n, k, seed, out);
        else hipLaunchKernelGGL((lean_serialize_kernel<uint32_t, false>), g, dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, n, k, seed, out);
    } else {
        if (big_endian) hipLaunchKernelGGL((lean_serialize_kernel<uint64_t, true>), g, dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n, k, seed, out);
        else hipLaunchKernelGGL((lean_serialize_kernel<uint64_t, false>), g, dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, n, k, seed, out);
    }
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_lean_deserialize(mhx_ctx *ctx, const uint8_t *d_records, int64_t n, int32_t k, int big_endian, int sig_dtype, void *d_sig,
                            int64_t *d_seeds, unsigned int *d_bad) {
    const dim3 g = row_grid(ctx, n);
    const uint32_t *in = (const uint32_t *)d_records;
    if (sig_dtype == MHX_U32) {
        if (big_endian) hipLaunchKernelGGL((lean_deserialize_kernel<uint32_t, true>), g, dim3(256), 0, ctx->stream, in, n, k, (uint32_t *)d_sig, d_seeds, d_bad);
        else hipLaunchKernelGGL((lean_deserialize_kernel<uint32_t, false>), g, dim3(256), 0, ctx->stream, in, n, k, (uint32_t *)d_sig, d_seeds, d_bad);
    } else {
        if (big_endian) hipLaunchKernelGGL((lean_deserialize_kernel<uint64_t, true>), g, dim3(256), 0, ctx->stream, in, n, k, (uint64_t *)d_sig, d_seeds, d_bad);
        else hipLaunchKernelGGL((lean_deserialize_kernel<uint64_t, false>), g, dim3(256), 0, ctx->stream, in, n, k, (uint64_t *)d_sig, d_seeds, d_bad);
    }
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_bbit_unpack(mhx_ctx *ctx, const uint64_t *d_blocks, int64_t n, int32_t k, int32_t b, uint32_t *d_out) {
    const int slot = bbit_slot_size(b);
    const int per = 64 / slot;
    const int nb = (k + per - 1) / per;
    const int64_t total = n * (int64_t)((k + 3) / 4);
    const dim3 g((unsigned)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, (int64_t)ctx->num_cus * 32)));
    if (k % 4 == 0 && ((uintptr_t)d_out & 15) == 0)
        hipLaunchKernelGGL(bbit_unpack_kernel<true>, g, dim3(256), 0, ctx->stream, d_blocks, n, k, slot, nb, d_out);
    else
        hipLaunchKernelGGL(bbit_unpack_kernel<false>, g, dim3(256), 0, ctx->stream, d_blocks, n, k, slot, nb, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx

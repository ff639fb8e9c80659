// band_digest.h -- FNV-1a-64 of one band key, shared by the digest kernel (pack_kernels.hip) and the LSH sort's key kernel
// (lsh_kernels.hip).  Product code: nothing here may reference oracle/.
#pragma once

#include <cstdint>
#include <type_traits>

#include <hip/hip_runtime.h>

namespace mhx {

// FNV-1a-64 of the band key of band `band` of row `row`: exactly the bytes the reference uses as that band's dictionary
// key (ref: datasketch/lsh.py:199,344,537-538: the r hashvalues of the band, each as 8 big-endian bytes) -- what
// MinHashLSH(hashfunc=fnv1a_64) would store (ref: lsh.py:540-543).
// a "signature matrix" that already holds the digests ([n, bands]: what band_digest_kernel wrote): the LSH sort then reads 8
// bytes per (row, band) instead of hashing r values again (config 3 computes the digests once)
struct Digest64 {
    uint64_t v;
};

template <typename SigT>
__device__ __forceinline__ uint64_t band_digest_of(const SigT *__restrict__ sig, int64_t row, int band, int32_t k, int32_t r) {
    if constexpr (std::is_same<SigT, Digest64>::value) {
        return sig[row * k + band].v;  // (k = bands here)
    } else {
    constexpr uint64_t kPrime = 0x100000001b3ull;
    constexpr uint64_t kPrime4 = kPrime * kPrime * kPrime * kPrime;  // four zero bytes: h ^= 0 leaves h, so h *= prime^4
    const SigT *src = sig + row * k + (int64_t)band * r;
    uint64_t h = 0xcbf29ce484222325ull;
    const auto absorb = [&](uint64_t v) {
        const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
        if (hi == 0) {  // every real hashvalue: the 4 leading key bytes are zero
            h *= kPrime4;
        } else {
#pragma unroll
            for (int byte = 3; byte >= 0; --byte) {
                h ^= (hi >> (8 * byte)) & 0xFFu;
                h *= kPrime;
            }
        }
#pragma unroll
        for (int byte = 3; byte >= 0; --byte) {  // big-endian byte order of the key
            h ^= (lo >> (8 * byte)) & 0xFFu;
            h *= kPrime;
        }
    };
    if constexpr (sizeof(SigT) == 4) {
        // uint32 signatures (the all-gather's wire format): the key bytes are those of the widened value
        if (((r | k) & 3) == 0 && (reinterpret_cast<uintptr_t>(sig) & 15) == 0) {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
            for (int c = 0; c < r / 4; ++c) {
                const uint4 v = src4[c];
                absorb(v.x);
                absorb(v.y);
                absorb(v.z);
                absorb(v.w);
            }
        } else {
            for (int c = 0; c < r; ++c) absorb(src[c]);
        }
    } else if (((r | k) & 1) == 0 && (reinterpret_cast<uintptr_t>(sig) & 15) == 0) {
        // 16-byte loads: a lane's band is r*8 contiguous bytes, but neighbouring lanes are r*8 bytes
        // apart, so every load instruction touches many lines -- fewer, wider loads it is
        const ulonglong2 *src2 = reinterpret_cast<const ulonglong2 *>(src);
        for (int c = 0; c < r / 2; ++c) {
            const ulonglong2 v = src2[c];
            absorb(v.x);
            absorb(v.y);
        }
    } else {
        for (int c = 0; c < r; ++c) absorb((uint64_t)src[c]);
    }
    return h;
    }
}

}  // namespace mhx

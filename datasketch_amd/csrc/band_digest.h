// band_digest.h -- FNV-1a-64 of one band key, shared by the digest kernel (pack_kernels.hip) and the LSH sort's key kernel
// (lsh_kernels.hip).  Product code: nothing here may reference oracle/.
#pragma once

#include <cstdint>
#include <type_traits>

#include <hip/hip_runtime.h>

namespace mhx {

// ---- FNV-1a-64 on a (hi, lo) pair of 32-bit registers, hand-scheduled for gfx950 ---------------------------------------
// h = (h ^ byte) * P with P = 2^40 + 0x1b3.  The byte only touches lo, so with x = lo ^ byte:
//   lo' = low32(x * 0x1b3),   hi' = low32(hi * 0x1b3 + (x << 8) + high32(x * 0x1b3)).
// Four instructions per byte: v_xor_b32_sdwa (the byte select rides on the xor), v_mad_u64_u32 (x * 0x1b3, both halves),
// v_lshl_add_u32, and a second v_mad_u64_u32 whose low half is hi * 0x1b3 + addend.  The compiler's own selection of
// the plain C++ form needs six (no SDWA for the middle bytes, v_mul_lo + v_lshlrev + v_add3 for the high word): the
// digest kernels are VALU-issue-bound (profiles/r05_pmc_sort_and_fused_before.txt: 150.7M VALU instructions per 40M
// digests = 0.28 of the fused kernel's 0.34 ms), so two instructions per byte are 15 % of the kernel.
template <int BYTE>
__device__ __forceinline__ uint32_t fnv_xor_byte(uint32_t lo, uint32_t w) {
    uint32_t x;
    if constexpr (BYTE == 3) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(x) : "v"(lo), "v"(w));
    else if constexpr (BYTE == 2) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(x) : "v"(lo), "v"(w));
    else if constexpr (BYTE == 1) asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(x) : "v"(lo), "v"(w));
    else asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(x) : "v"(lo), "v"(w));
    return x;
}

// low 32 bits of a * c + add (v_mad_u64_u32: the 64-bit addend's high half is zero and never looked at)
__device__ __forceinline__ uint32_t mad_lo32(uint32_t a, uint32_t c, uint32_t add) {
    uint64_t r;
    const uint64_t add64 = add;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c), "v"(add64) : "vcc");
    return (uint32_t)r;
}

__device__ __forceinline__ void fnv_times_prime(uint32_t &hi, uint32_t &lo, uint32_t x) {  // (hi:x) * P mod 2^64
    const uint64_t t = (uint64_t)x * 0x1b3u;
    hi = mad_lo32(hi, 0x1b3u, (x << 8) + (uint32_t)(t >> 32));
    lo = (uint32_t)t;
}

// the eight big-endian key bytes of one hashvalue (ref: datasketch/lsh.py:537-538), hi word first
__device__ __forceinline__ void fnv_absorb_value(uint32_t &hi, uint32_t &lo, uint32_t vhi, uint32_t vlo) {
    constexpr uint64_t kPrime = 0x100000001b3ull;
    constexpr uint64_t kPrime4 = kPrime * kPrime * kPrime * kPrime;  // four zero bytes: h ^= 0 leaves h, so h *= prime^4
    if (vhi == 0) {  // every real hashvalue (< 2^32): three multiply-adds
        const uint64_t t = (uint64_t)lo * (uint32_t)kPrime4;
        hi = mad_lo32(hi, (uint32_t)kPrime4, mad_lo32(lo, (uint32_t)(kPrime4 >> 32), (uint32_t)(t >> 32)));
        lo = (uint32_t)t;
    } else {
        fnv_times_prime(hi, lo, fnv_xor_byte<3>(lo, vhi));
        fnv_times_prime(hi, lo, fnv_xor_byte<2>(lo, vhi));
        fnv_times_prime(hi, lo, fnv_xor_byte<1>(lo, vhi));
        fnv_times_prime(hi, lo, fnv_xor_byte<0>(lo, vhi));
    }
    fnv_times_prime(hi, lo, fnv_xor_byte<3>(lo, vlo));
    fnv_times_prime(hi, lo, fnv_xor_byte<2>(lo, vlo));
    fnv_times_prime(hi, lo, fnv_xor_byte<1>(lo, vlo));
    fnv_times_prime(hi, lo, fnv_xor_byte<0>(lo, vlo));
}

// FNV-1a-64 of the band key of band `band` of row `row`: exactly the bytes the reference uses as that band's dictionary
// key (ref: datasketch/lsh.py:199,344,537-538: the r hashvalues of the band, each as 8 big-endian bytes) -- what
// MinHashLSH(hashfunc=fnv1a_64) would store (ref: lsh.py:540-543).
// a "signature matrix" that already holds the digests ([n, bands]: what band_digest_kernel wrote): the LSH sort then reads 8
// bytes per (row, band) instead of hashing r values again (config 3 computes the digests once)
struct Digest64 {
    uint64_t v;
};
// ... the same digests band-major ([bands, n]: band j's digests of all rows are contiguous) -- the layout the bucketing reads
// with unit stride (a team takes 2048 rows of ONE band) and the per-band hashtables of the reference suggest (lsh.py:199)
struct Digest64BM {
    uint64_t v;
};

template <typename SigT>
__device__ __forceinline__ uint64_t band_digest_of(const SigT *__restrict__ sig, int64_t row, int band, int32_t k, int32_t r, int64_t n = 0) {
    if constexpr (std::is_same<SigT, Digest64>::value) {
        return sig[row * k + band].v;  // (k = bands here)
    } else if constexpr (std::is_same<SigT, Digest64BM>::value) {
        return sig[(int64_t)band * n + row].v;
    } else {
    const SigT *src = sig + row * k + (int64_t)band * r;
    uint32_t h_hi = 0xcbf29ce4u, h_lo = 0x84222325u;
    const auto absorb = [&](uint64_t v) { fnv_absorb_value(h_hi, h_lo, (uint32_t)(v >> 32), (uint32_t)v); };
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
    return ((uint64_t)h_hi << 32) | h_lo;
    }
}

}  // namespace mhx

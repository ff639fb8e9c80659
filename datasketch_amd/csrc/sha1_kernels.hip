// sha1_kernels.hip -- the reference's default token hash on the device.
//
// Reference: datasketch/hashfunc.py:5-15   sha1_hash32(data) = first 4 bytes of SHA1(data), little-endian
//            datasketch/hashfunc.py:17-28  sha1_hash64(data) = first 8 bytes, little-endian
// called once per token from MinHash.update / update_batch (datasketch/minhash.py:221, :262-263).
// In Python that per-token call is >100x the cost of the permutation kernel; here a corpus of byte
// tokens (one packed byte buffer + int64 offsets, the CSR of bytes) is hashed by one thread per
// token and the uint32 / uint64 results feed mhx_minhash_bulk_dev directly.
//
// SHA-1 (FIPS 180-4) with a 16-word circular message schedule in registers; 80 rounds fully
// unrolled.  Tokens are read as aligned dwords and re-aligned with v_alignbyte (a token may start at
// any byte); an aligned dword is only loaded if it holds at least one byte of the token, so the
// kernel never touches memory past the packed buffer's last dword.
#include "mhx_internal.h"

namespace mhx {
namespace {

__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return __builtin_rotateleft32(x, n); }

// little-endian dword k (bytes 4k..4k+3) of the token's byte stream; bytes at or past `len` are 0
__device__ __forceinline__ uint32_t stream_word(const uint32_t *__restrict__ aligned, uint32_t shift, int64_t len,
                                                int64_t k) {
    const int64_t first = 4 * k;  // stream position of this word's first byte
    if (first >= len) return 0;
    // aligned dwords j = k and k+1 cover stream bytes [4j - shift, 4j - shift + 4)
    const uint32_t lo = aligned[k];
    const bool need_hi = shift != 0 && (4 * (k + 1) - (int64_t)shift) < len;
    const uint32_t hi = need_hi ? aligned[k + 1] : 0u;
    uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, shift);
    const int64_t valid = len - first;  // > 0
    if (valid < 4) w &= (1u << (8 * (uint32_t)valid)) - 1u;
    return w;
}

// message word k (big-endian, as SHA-1 consumes it) of the PADDED message: data, 0x80, zeros, bit length
__device__ __forceinline__ uint32_t padded_word(const uint32_t *__restrict__ aligned, uint32_t shift, int64_t len,
                                                int64_t k, int64_t total_words) {
    uint32_t w = stream_word(aligned, shift, len, k);
    const int64_t first = 4 * k;
    if (len >= first && len < first + 4) w |= 0x80u << (8 * (uint32_t)(len - first));
    w = __builtin_bswap32(w);
    const uint64_t bits = (uint64_t)len * 8u;
    if (k == total_words - 2) w = (uint32_t)(bits >> 32);
    if (k == total_words - 1) w = (uint32_t)bits;
    return w;
}

template <typename OutT>
__global__ __launch_bounds__(256) void sha1_tokens_kernel(const uint8_t *__restrict__ bytes,
                                                          const int64_t *__restrict__ offsets, int64_t n_tokens,
                                                          OutT *__restrict__ out) {
    for (int64_t tok = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tok < n_tokens;
         tok += (int64_t)gridDim.x * blockDim.x) {
        const int64_t beg = offsets[tok];
        const int64_t len = offsets[tok + 1] - beg;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(bytes) + (uintptr_t)beg;
        const uint32_t shift = (uint32_t)(addr & 3u);
        const uint32_t *aligned = reinterpret_cast<const uint32_t *>(addr - shift);
        const int64_t n_blocks = (len + 8) / 64 + 1;
        const int64_t total_words = n_blocks * 16;
        uint32_t h0 = 0x67452301u, h1 = 0xEFCDAB89u, h2 = 0x98BADCFEu, h3 = 0x10325476u, h4 = 0xC3D2E1F0u;
        for (int64_t blk = 0; blk < n_blocks; ++blk) {
            uint32_t w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = padded_word(aligned, shift, len, blk * 16 + i, total_words);
            uint32_t a = h0, b = h1, c = h2, d = h3, e = h4;
#pragma unroll
            for (int t = 0; t < 80; ++t) {
                uint32_t wt;
                if (t < 16) {
                    wt = w[t];
                } else {
                    wt = rotl(w[(t - 3) & 15] ^ w[(t - 8) & 15] ^ w[(t - 14) & 15] ^ w[t & 15], 1);
                    w[t & 15] = wt;
                }
                uint32_t f, kc;
                if (t < 20) {
                    f = (b & c) | (~b & d);
                    kc = 0x5A827999u;
                } else if (t < 40) {
                    f = b ^ c ^ d;
                    kc = 0x6ED9EBA1u;
                } else if (t < 60) {
                    f = (b & c) | (b & d) | (c & d);
                    kc = 0x8F1BBCDCu;
                } else {
                    f = b ^ c ^ d;
                    kc = 0xCA62C1D6u;
                }
                const uint32_t tmp = rotl(a, 5) + f + e + kc + wt;
                e = d;
                d = c;
                c = rotl(b, 30);
                b = a;
                a = tmp;
            }
            h0 += a;
            h1 += b;
            h2 += c;
            h3 += d;
            h4 += e;
        }
        // digest bytes are h0, h1, ... big-endian; the reference unpacks the first 4 / 8 of them "<I" / "<Q"
        const uint32_t lo = __builtin_bswap32(h0);
        if (sizeof(OutT) == 4)
            out[tok] = (OutT)lo;
        else
            out[tok] = (OutT)(((uint64_t)__builtin_bswap32(h1) << 32) | lo);
    }
}

}  // namespace

int launch_sha1_tokens(mhx_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n_tokens,
                       int out_dtype, void *d_out) {
    if (n_tokens == 0) return MHX_OK;
    const int64_t want = (n_tokens + 255) / 256;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 32));
    if (out_dtype == MHX_U32)
        hipLaunchKernelGGL(sha1_tokens_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream, d_bytes, d_offsets,
                           n_tokens, static_cast<uint32_t *>(d_out));
    else if (out_dtype == MHX_U64)
        hipLaunchKernelGGL(sha1_tokens_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream, d_bytes, d_offsets,
                           n_tokens, static_cast<uint64_t *>(d_out));
    else
        return fail(MHX_ERR_INVALID, "unknown out_dtype %d", out_dtype);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx

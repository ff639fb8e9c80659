// minhash_kernels.hip -- bulk MinHash signatures for gfx950 (MI355X), hand-written HIP.
//
// What is computed (reference: datasketch/minhash.py:293-297 inside the per-set loop of
// :491-522):   out[i,k] = min(init[i,k], min_t fold((hv[t]*a[k] + b[k]) mod 2^64))
// with fold(s) = (s mod (2^61-1)) & 0xFFFFFFFF.  numpy evaluates hv*a+b in uint64, so the
// 2^64 wrap is part of the function and is reproduced here.
//
// Design (see DESIGN.md "MinHash kernel"):
//   * permutations on lanes: lane l of a wave keeps (a,b) of P permutations in VGPRs for the
//     whole kernel; one wave walks one set; the running minima stay in registers, so there is
//     no cross-lane reduction at all and the [K] row is stored fully coalesced.
//   * tokens are wave-uniform: they are fetched with scalar loads (s_load_dwordx16 = 8 uint64
//     tokens) into SGPRs and feed v_mad_u64_u32 directly as the scalar operand; the next chunk
//     is prefetched while the current one is being hashed.
//   * hashing a (token, permutation) pair in full is integer VALU: two 32x32 multiplies and the
//     Mersenne fold ("full evaluation": a 3-op fold whose rare failure, probability 2^-29 per pair,
//     shows in the final minima; such a set is recomputed with the exact fold).
//   * the default path does not hash most pairs in full: a "sieve" keeps, per row of 16 tokens, the
//     minimum of the LOW WORD of the hash (one multiply per pair, one v_min3 per two pairs), which pins
//     down the one token that can hold the minimum AND proves that no other token is close enough to
//     matter; only that token is hashed exactly (rescan of the best row from a wave-private LDS tile).
//     Results are bit-exact unconditionally: a set whose proof fails is flagged and a second launch
//     settles it (repeated tokens dropped + sieve, else the full evaluation).
//   * sets with few, long token lists (update_batch on one MinHash) are split over many waves
//     and combined with 64-bit atomic min.
// The kernels are VALU-bound (about 75 integer ops per input byte), not HBM-bound.
#include <type_traits>

#include "mhx_internal.h"

namespace mhx {
namespace {

constexpr uint64_t kMersenne = (1ull << 61) - 1;  // datasketch/minhash.py:30
constexpr uint32_t kMaxHash = 0xFFFFFFFFu;        // datasketch/minhash.py:31
constexpr int kWave = 64;

struct BulkArgs {
    const void *hv;          // tokens (uint64 or uint32)
    const int64_t *offsets;  // CSR or nullptr
    int64_t fixed_len;
    int64_t n_sets;
    const uint64_t *a;
    const uint64_t *b;
    int32_t num_perm;
    int32_t path;            // 0 sieve (+ fallbacks), 1 exact fold everywhere, 2 fast fold (+ exact redo)
    unsigned long long *stats;  // optional device counters: [0] sets the sieve launch left to the full one,
                                // [1] sets redone with the exact fold, [2] sieve blocks, [3] sets hashed pair by pair
    uint8_t *redo;              // wave-per-set kernels: redo[set] == 1 <=> the sieve launch left the set to the dedup launch,
                                // == 2 <=> the dedup launch left it to the pairwise (full) launch
    int32_t redo_match;         // the flag value this launch works on (MODE_DEDUP: 1, MODE_FULL after it: 2)
    unsigned int *sieve_hint;   // sieve launch: [0] sets tried, [1] proofs failed so far in this launch (zeroed before it)
    unsigned int *mode_word;    // what the previous call on this context learned about the corpus: 1 = most sets defeat the
                                // one-candidate proof (repeated tokens), so the first launch is the tie-tolerant one (MODE_SIEVE_TIES)
    unsigned int *pair_count;   // flagged launches: [0] sets the dedup launch left to the pairwise one; the first kPairListCap of
    unsigned int *pair_list;    // them are listed here (set numbers), so that the pairwise launch need not scan the flags for a handful
    int32_t prefetch;        // warm the next set's tokens with a vector load (option minhash.prefetch)
    int32_t ties;            // second launch: try the tie-tolerant sieve before the dedup pass (option minhash.ties)
    int32_t share_last;      // 3 or 4 permutations per lane: lane groups share the rows of a partly filled last slot (option minhash.share)
    int64_t alias_mask;      // profiling only (option minhash.alias): sets read tokens of set (i & mask); -1 = off
    const uint64_t *init;
    int64_t init_stride;
    void *out;
};

// ---- per-pair arithmetic ------------------------------------------------------------------
// s = (h*a + b) mod 2^64, h < 2^32:   low 64 bits of h*a_lo + b, plus (h*a_hi mod 2^32) << 32
__device__ __forceinline__ void mad_narrow(uint32_t h, uint32_t a_lo, uint32_t a_hi, uint64_t b,
                                           uint32_t &s_lo, uint32_t &s_hi) {
    const uint64_t s0 = (uint64_t)h * a_lo + b;  // v_mad_u64_u32 (wraps mod 2^64 like numpy)
    s_lo = (uint32_t)s0;
    s_hi = (uint32_t)(s0 >> 32) + h * a_hi;  // v_mul_lo_u32 + v_add_u32
}

// general uint64 token (sha1_hash64, identity on big ints): one more 32-bit multiply
__device__ __forceinline__ void mad_wide(uint32_t h_lo, uint32_t h_hi, uint32_t a_lo, uint32_t a_hi,
                                         uint64_t b, uint32_t &s_lo, uint32_t &s_hi) {
    const uint64_t s0 = (uint64_t)h_lo * a_lo + b;
    s_lo = (uint32_t)s0;
    s_hi = (uint32_t)(s0 >> 32) + h_lo * a_hi + h_hi * a_lo;
}

// Exact fold: (s mod p) & 0xFFFFFFFF for p = 2^61-1.
//   s = top*2^61 + low  ==  top + low (mod p), and top + low < 2p, so
//   s mod p = y - p*[y >= p] with y = low + top;  -p == +1 (mod 2^32).
__device__ __forceinline__ uint32_t fold_exact(uint32_t s_lo, uint32_t s_hi) {
    const uint32_t top = s_hi >> 29;
    const uint64_t y = ((((uint64_t)(s_hi & 0x1FFFFFFFu)) << 32) | s_lo) + top;
    return (uint32_t)y + (y >= kMersenne ? 1u : 0u);
}

// Fast fold.  The fast path hashes with b' = b + 1 (mod 2^64), i.e. it sees s' = s + 1, and
// computes u = s'_lo + (s'_hi >> 29) (mod 2^32): two full-rate VALU ops.
//   * s_lo != 2^32-1: s'_lo = s_lo + 1 without carry and s'_hi = s_hi, so u = s_lo + top + 1.
//     If u >= 8 that sum did not wrap, hence low61(s) + top < p, no "-p" correction is due and
//     fold_exact(s) == u - 1.
//   * s_lo == 2^32-1: u = (s_hi + 1) >> 29 <= 7.
// So u <= 7 is the only way the exact result can differ from u - 1 (or the order of two values
// can flip); probability 2^-29 per pair.  A set whose final minimum of u is <= 7 for some
// permutation is recomputed with fold_exact; otherwise min(fold_exact) == min(u) - 1.
__device__ __forceinline__ uint32_t fold_fast(uint32_t s_lo, uint32_t s_hi) {
    return s_lo + (s_hi >> 29);
}

template <bool EXACT>
__device__ __forceinline__ uint32_t fold(uint32_t s_lo, uint32_t s_hi) {
    return EXACT ? fold_exact(s_lo, s_hi) : fold_fast(s_lo, s_hi);
}

template <int P>
struct Perms {
    uint32_t a_lo[P], a_hi[P];
    uint64_t b[P];
};

// ---- token chunks in SGPRs ----------------------------------------------------------------
// Tokens are read through the constant address space: a wave-uniform load from it is always
// selected as a scalar load (s_load_dwordxN -> SGPRs, scalar cache), which costs no VALU issue
// and lets the token feed v_mad_u64_u32 as its scalar operand.  The corpus is read-only for the
// whole launch, which is what the scalar cache requires.
#define MHX_CONST_AS __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const T MHX_CONST_AS *as_const(const T *p) {
    return (const T MHX_CONST_AS *)p;  // NOLINT: address-space cast
}

template <typename TokT>
struct Chunk;
template <>
struct Chunk<uint64_t> {
    static constexpr int N = 8;  // 64 B = one s_load_dwordx16
    uint32_t w[2 * N];           // little-endian halves: token i = (w[2i+1] << 32) | w[2i]
    __device__ __forceinline__ void load(const uint64_t MHX_CONST_AS *p) {
        const uint32_t MHX_CONST_AS *q = (const uint32_t MHX_CONST_AS *)p;
#pragma unroll
        for (int i = 0; i < 2 * N; ++i) w[i] = q[i];
    }
    __device__ __forceinline__ uint32_t or_hi() const {  // 7 x s_or_b32 on the scalar unit
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= w[2 * i + 1];
        return o;
    }
    __device__ __forceinline__ uint32_t lo(int i) const { return w[2 * i]; }
    __device__ __forceinline__ uint32_t hi(int i) const { return w[2 * i + 1]; }
    // the sieve reads low words only; this keeps the fetch one s_load_dwordx16 instead of 8 s_load_dword
    __device__ __forceinline__ void keep_whole() const {
        asm volatile("" ::"s"(w[1]), "s"(w[3]), "s"(w[5]), "s"(w[7]), "s"(w[9]), "s"(w[11]), "s"(w[13]), "s"(w[15]));
    }
    // "the chunk has arrived": scalar loads return out of order, so the only wait is lgkmcnt(0); a use
    // placed BEFORE the next prefetch is issued makes that wait cover this chunk alone
    __device__ __forceinline__ void arrived() const { asm volatile("" ::"s"(w[0])); }
};
template <>
struct Chunk<uint32_t> {
    static constexpr int N = 16;
    uint32_t v[N];
    __device__ __forceinline__ void load(const uint32_t MHX_CONST_AS *p) {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
    __device__ __forceinline__ uint32_t or_hi() const { return 0; }
    __device__ __forceinline__ uint32_t lo(int i) const { return v[i]; }
    __device__ __forceinline__ uint32_t hi(int) const { return 0; }
    __device__ __forceinline__ void keep_whole() const {}
    __device__ __forceinline__ void arrived() const { asm volatile("" ::"s"(v[0])); }
};

template <int P, bool EXACT, bool WIDE, typename TokT>
__device__ __forceinline__ void hash_chunk(const Chunk<TokT> &c, const Perms<P> &pm,
                                           uint32_t (&acc)[P]) {
    constexpr int N = Chunk<TokT>::N;
    if (!WIDE) {
#pragma unroll
        for (int i = 0; i < N; i += 4) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                uint32_t f[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t lo, hi;
                    mad_narrow(c.lo(i + j), pm.a_lo[p], pm.a_hi[p], pm.b[p], lo, hi);
                    f[j] = fold<EXACT>(lo, hi);
                }
                acc[p] = min(min(acc[p], f[0]), f[1]);  // v_min3_u32
                acc[p] = min(min(acc[p], f[2]), f[3]);  // v_min3_u32
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                uint32_t l0, h0;
                mad_wide(c.lo(i), c.hi(i), pm.a_lo[p], pm.a_hi[p], pm.b[p], l0, h0);
                acc[p] = min(acc[p], fold<EXACT>(l0, h0));
            }
        }
    }
}

// Hash tokens [beg, end) of one set into acc (fold<EXACT> space).  All arguments wave-uniform.
// Two loops: the narrow one (every token < 2^32: two multiplies per pair) runs until a chunk with
// a wider token shows up, the wide one (three multiplies) finishes the set.  Keeping them apart
// keeps the hot loop free of branches.  Each iteration issues the scalar load of the NEXT chunk
// first and only then hashes the current one; the s_waitcnt for the prefetched SGPRs therefore
// sits a whole chunk of VALU work (about 100 instructions) after the s_load.
template <int P, bool EXACT, typename TokT>
__device__ __forceinline__ void hash_range(const TokT MHX_CONST_AS *hv, int64_t beg, int64_t end,
                                           const Perms<P> &pm, uint32_t (&acc)[P]) {
    constexpr int N = Chunk<TokT>::N;
    const TokT MHX_CONST_AS *p = hv + beg;
    const int64_t n = end - beg;
    const int nfull = (int)(n / N);  // launcher keeps per-wave ranges far below 2^31 chunks
    if (nfull > 0) {
        const auto chunk_ptr = [&](int idx) { return p + (int64_t)(idx < nfull ? idx : nfull - 1) * N; };
        // Narrow loop over PAIRS of chunks with two ping-pong buffers: no SGPR copies, and every
        // load is consumed on the fall-through path, so the optimiser cannot sink it below the math.
        Chunk<TokT> a, b;
        a.load(p);
        const int npairs = nfull >> 1;
        int i = 0;  // chunk index held by `a`
        bool wide = false;
        for (int j = 0; j < npairs; ++j) {
            if (a.or_hi() != 0) {
                wide = true;
                break;
            }
            b.load(chunk_ptr(i + 1));           // prefetch (always a real chunk)
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the math
            hash_chunk<P, EXACT, false, TokT>(a, pm, acc);
            ++i;
            if (b.or_hi() != 0) {
                a = b;
                wide = true;
                break;
            }
            a.load(chunk_ptr(i + 1));           // next pair's first chunk (clamped at the end)
            __builtin_amdgcn_sched_barrier(0);
            hash_chunk<P, EXACT, false, TokT>(b, pm, acc);
            ++i;
        }
        if (!wide && (nfull & 1)) {  // odd chunk left in `a`
            if (a.or_hi() != 0) {
                wide = true;
            } else {
                hash_chunk<P, EXACT, false, TokT>(a, pm, acc);
                ++i;
            }
        }
        if (wide) {  // `a` holds chunk i < nfull; rare path, plain double buffering
            for (;;) {
                b.load(chunk_ptr(i + 1));
                __builtin_amdgcn_sched_barrier(0);
                hash_chunk<P, EXACT, true, TokT>(a, pm, acc);
                a = b;
                if (++i >= nfull) break;
            }
        }
    }
    for (int64_t t = (int64_t)nfull * N; t < n; ++t) {  // ragged tail, one token at a time
        const uint64_t tok = p[t];
        const uint32_t lo = (uint32_t)tok, hi = (uint32_t)(tok >> 32);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            uint32_t l0, h0;
            mad_wide(lo, hi, pm.a_lo[q], pm.a_hi[q], pm.b[q], l0, h0);
            acc[q] = min(acc[q], fold<EXACT>(l0, h0));
        }
    }
}

// ---- sieve: find the minimum without hashing every pair ------------------------------------
// The exact value of a pair is R = fold_exact(s) = s_lo + top + ge (mod 2^32) with top = s >> 61
// <= 7 and ge <= 1, and s_lo = lo32(h_lo * a_lo + b_lo) needs ONE multiply, whatever the high
// words of h, a and b are.  With the key  M = lo32(h_lo*a_lo + b_lo + 8)  (bias 8, see below):
//     R + 8 = M + e,  0 <= e <= 8,  without wrap-around, for EVERY token of a block whose
//     smallest key is >= 16  (no token then has s_lo in [2^32-8, 2^32), the only place where
//     s_lo + top + ge can wrap).
// So the token with the smallest R is among the tokens with M <= min M + 8.  A block of up to 256
// tokens is cut into rows of 16 consecutive tokens and the hot loop only keeps min M per row: one
// v_mad_u64_u32 per pair and one v_min3_u32 per TWO pairs, instead of two multiplies, shift, add
// and half a min3.  Finished rows are tagged with their index in the low 4 bits (key & ~15 | row)
// and folded into the two smallest tagged values k1 < k2; k2 - k1 >= 32 proves that every token
// outside the best row has M >= min M + 17.  The best row is then rescanned (16 keys, the same
// k1/k2 fold over the column index): if its two smallest keys are also >= 32 apart, (best row,
// best column) is the ONLY token with M <= min M + 16: it alone is hashed exactly.  The rescan
// reads per-lane addresses (every lane has its own best row), which the vector memory path serves
// at one lane per clock -- 34 such loads per set made the kernel texture-address bound (3.7 ms).
// So each block is also copied once, coalesced, into a wave-private LDS tile (rows padded to
// distinct banks) and the rescan reads LDS.  If a proof fails in any lane (equal tokens at the minimum, two keys within 32, min
// key < 16; about 4 sets in 10^4 of distinct random tokens) the whole set is redone by the full
// evaluation, so results stay bit-exact unconditionally.
// (Round-1 history: the first sieve kept row AND column minima in the hot loop -- two half min3 per
// pair -- to pin the cell down without a rescan; the rescan costs 16 keys per block instead of 256
// column updates and frees 32 VGPRs.)
constexpr int kRowTokens = 16;
constexpr int kBlockRows = 16;  // 256 tokens
// LDS tile of one block: row stride in dwords.  144 B (uint64 tokens) / 80 B (uint32): multiples of
// 16 B for ds_write_b128, and the 16 row starts fall into 16 different banks (36*r and 20*r mod 64).
template <typename TokT>
struct StageLayout {
    static constexpr int kStride = sizeof(TokT) == 8 ? 36 : 20;
    static constexpr int kWordsPerTok = sizeof(TokT) / 4;
};
constexpr int kStageWordsPerWave = kBlockRows * 36;

__device__ __forceinline__ uint32_t sieve_key(uint32_t h, uint32_t a_lo, uint64_t b8) {
    // Only the low word is used, but one v_mad_u64_u32 beats v_mul_lo_u32 + v_add_u32; the empty
    // asm keeps the optimiser from narrowing the 64-bit multiply-add.
    uint64_t r = (uint64_t)h * a_lo + b8;
    asm("" : "+v"(r));
    return (uint32_t)r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t x, uint32_t y, uint32_t z) { return min(min(x, y), z); }
__device__ __forceinline__ uint32_t tag16(uint32_t key, uint32_t idx) { return (key & ~15u) | idx; }  // v_and_or_b32

template <int P>
struct SievePerms {
    uint32_t a_lo[P];
    uint64_t b8[P];  // b + 8 (only the low word matters)
    bool active[P];  // lane holds a real permutation (k < num_perm)
    // (wave-uniform) lanes per group in the last slot: 64 = every lane its own permutation.  When the last slot holds r <= 32
    // permutations (K = 136: 8 of 64 lanes busy in the third slot), 64/span lane groups hold the SAME r permutations and
    // take every (64/span)-th row of a block each -- see sieve_range
    uint32_t span = kWave;
};

// (smallest, second smallest) of the tagged keys seen so far
struct Two {
    uint32_t k1 = kMaxHash, k2 = kMaxHash;
    __device__ __forceinline__ void add(uint32_t key) {
        k2 = umed3(k1, k2, key);
        k1 = min(k1, key);
    }
    __device__ __forceinline__ bool apart() const { return k2 - k1 >= 32u; }  // k2 >= k1 always
    __device__ __forceinline__ void merge_from_lane(int other) {  // this record and lane `other`'s (over disjoint rows)
        const uint32_t o1 = (uint32_t)__shfl((int)k1, other, kWave), o2 = (uint32_t)__shfl((int)k2, other, kWave);
        add(o1);
        add(o2);
    }
};

// the three smallest: what the tie-tolerant proof of the second launch needs (see finish_block_ties)
struct Three {
    uint32_t k1 = kMaxHash, k2 = kMaxHash, k3 = kMaxHash;
    __device__ __forceinline__ void add(uint32_t key) {
        k3 = umed3(k2, k3, key);
        k2 = umed3(k1, k2, key);
        k1 = min(k1, key);
    }
    __device__ __forceinline__ void merge_from_lane(int other) {
        const uint32_t o1 = (uint32_t)__shfl((int)k1, other, kWave), o2 = (uint32_t)__shfl((int)k2, other, kWave);
        const uint32_t o3 = (uint32_t)__shfl((int)k3, other, kWave);
        add(o1);
        add(o2);
        add(o3);
    }
};

// fold the keys of one chunk into the row minimum; OPEN: the chunk starts the row
template <int P, typename TokT, bool OPEN, int PS = P>  // PS: the slots taken here (P, or P - 1 when the last one is shared out)
__device__ __forceinline__ void sieve_chunk(const Chunk<TokT> &c, const SievePerms<P> &sp, uint32_t (&row)[P]) {
    constexpr int N = Chunk<TokT>::N;
    c.keep_whole();
#pragma unroll
    for (int i = 0; i < N; i += 4) {
#pragma unroll
        for (int p = 0; p < PS; ++p) {
            const uint32_t m0 = sieve_key(c.lo(i + 0), sp.a_lo[p], sp.b8[p]);
            const uint32_t m1 = sieve_key(c.lo(i + 1), sp.a_lo[p], sp.b8[p]);
            const uint32_t m2 = sieve_key(c.lo(i + 2), sp.a_lo[p], sp.b8[p]);
            const uint32_t m3 = sieve_key(c.lo(i + 3), sp.a_lo[p], sp.b8[p]);
            row[p] = (OPEN && i == 0) ? min(m0, m1) : umin3(row[p], m0, m1);
            row[p] = umin3(row[p], m2, m3);
        }
        // pin the minima here: left alone, the optimiser hoists every multiply of the chunk above
        // the first min3 and keeps all their results live (113 VGPRs instead of ~70)
        if constexpr (PS == 1)
            asm volatile("" : "+v"(row[0]));
        else
            asm volatile("" : "+v"(row[0]), "+v"(row[PS - 1]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// End of a block: rescan the best row of every permutation in the LDS tile (row stride STRIDE dwords,
// WPT dwords per token), prove uniqueness, hash the one candidate exactly.  Returns the lanes whose
// proof failed.  PARTIAL: row `last_row` holds only `last_cols` tokens (the dedup launch's compacted tile);
// the cells behind them are stale and their keys are replaced by 2^32-1.
template <int P, int STRIDE, int WPT, bool PARTIAL = false>
__device__ __forceinline__ bool finish_block(const Two (&rows)[P], const uint32_t *tile, const Perms<P> &pm,
                                             const SievePerms<P> &sp, uint32_t (&res)[P], uint32_t last_row = 16,
                                             uint32_t last_cols = 16) {
    Two cols[P];
    uint64_t best[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const uint32_t myrow = rows[q].k1 & 15u;
        const uint32_t *rowp = tile + myrow * STRIDE;  // per-lane LDS address
        const uint32_t limit = PARTIAL ? (myrow == last_row ? last_cols : 16u) : 16u;
#pragma unroll
        for (int c = 0; c < kRowTokens; ++c) {
            uint32_t m = (uint32_t)((uint64_t)rowp[c * WPT] * sp.a_lo[q] + sp.b8[q]);
            if (PARTIAL) m = (uint32_t)c < limit ? m : kMaxHash;
            cols[q].add(tag16(m, (uint32_t)c));
        }
        const uint32_t j1 = cols[q].k1 & 15u;
        best[q] = WPT == 2 ? *reinterpret_cast<const uint64_t *>(rowp + 2 * j1) : (uint64_t)rowp[j1];
        __builtin_amdgcn_sched_barrier(0);  // one permutation's 16 LDS words at a time (registers)
    }
    bool fail = false;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const bool ok = rows[q].apart() && cols[q].apart() && cols[q].k1 >= 16u;
        fail |= sp.active[q] && !ok;
        uint32_t l0, h0;
        mad_wide((uint32_t)best[q], (uint32_t)(best[q] >> 32), pm.a_lo[q], pm.a_hi[q], pm.b[q], l0, h0);
        res[q] = min(res[q], fold_exact(l0, h0));
    }
    return fail;
}

// The same for a set the first launch could not settle, with a proof that tolerates ONE tie.  With B = the best key
// rounded down to 16, a token is "in the window" when its tagged key is < B + 32; every token outside has a key
// >= B + 32 while the token with the smallest exact value has a key <= min key + 8 < B + 24 -- so the exact minimum
// is among the window's tokens, however many they are.  The first launch insists on one; here two are accepted
// (hash both, take the smaller): either two cells of the best row (third smallest key of the row outside the
// window) or the best cells of two rows (third smallest row outside; each of the two rows with a single cell in
// the window -- the second row is rescanned for that).  That settles a token that occurs twice -- what defeats
// the first launch on real token lists -- and two genuinely close keys, at ~1.2x the instructions of the plain sieve
// instead of the dedup pass's 1.55x.  Three or more in the window (a token occurring three times): the set goes on
// to the dedup pass.  (Keys near 2^32 are refused so that the 2^32-1 a fold starts from never looks like a tie.)
template <int P, int STRIDE, int WPT>
__device__ __forceinline__ bool finish_block_ties(const Three (&rows)[P], const uint32_t *tile, const Perms<P> &pm,
                                                  const SievePerms<P> &sp, uint32_t (&res)[P], bool *tied = nullptr) {
    bool fail = false;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const uint32_t base = rows[q].k1 & ~15u;
        const uint32_t *rowp = tile + (rows[q].k1 & 15u) * STRIDE;  // per-lane LDS address
        Three cols;
#pragma unroll
        for (int c = 0; c < kRowTokens; ++c) {
            const uint32_t m = (uint32_t)((uint64_t)rowp[c * WPT] * sp.a_lo[q] + sp.b8[q]);
            cols.add(tag16(m, (uint32_t)c));
        }
        const bool row_tie = rows[q].k2 - base < 32u;  // a second row reaches into the window
        const bool col_tie = cols.k2 - base < 32u;     // a second cell of the best row does
        bool ok = base >= 16u && base <= 0xFFFFFF00u && rows[q].k3 - base >= 32u && !(row_tie && col_tie) && cols.k3 - base >= 32u;
        const uint32_t j1 = cols.k1 & 15u;
        const uint64_t t1 = WPT == 2 ? *reinterpret_cast<const uint64_t *>(rowp + 2 * j1) : (uint64_t)rowp[j1];
        const uint32_t j2 = col_tie ? (cols.k2 & 15u) : j1;
        uint64_t t2 = WPT == 2 ? *reinterpret_cast<const uint64_t *>(rowp + 2 * j2) : (uint64_t)rowp[j2];
        if (__any(row_tie)) {  // wave-uniform: rescan the second row where there is one (the other lanes go through the motions)
            const uint32_t *row2 = tile + (rows[q].k2 & 15u) * STRIDE;
            Two other;
#pragma unroll
            for (int c = 0; c < kRowTokens; ++c) {
                const uint32_t m = (uint32_t)((uint64_t)row2[c * WPT] * sp.a_lo[q] + sp.b8[q]);
                other.add(tag16(m, (uint32_t)c));
            }
            const uint32_t i1 = other.k1 & 15u;
            const uint64_t u1 = WPT == 2 ? *reinterpret_cast<const uint64_t *>(row2 + 2 * i1) : (uint64_t)row2[i1];
            if (row_tie) {
                ok = ok && other.k2 - base >= 32u;
                t2 = u1;
            }
        }
        fail |= sp.active[q] && !ok;
        if (tied) *tied |= sp.active[q] && (row_tie || col_tie);  // the one-candidate proof would have failed here
        uint32_t l0, h0, l1, h1;
        mad_wide((uint32_t)t1, (uint32_t)(t1 >> 32), pm.a_lo[q], pm.a_hi[q], pm.b[q], l0, h0);
        mad_wide((uint32_t)t2, (uint32_t)(t2 >> 32), pm.a_lo[q], pm.a_hi[q], pm.b[q], l1, h1);
        res[q] = umin3(res[q], fold_exact(l0, h0), fold_exact(l1, h1));
        __builtin_amdgcn_sched_barrier(0);  // one permutation at a time (registers)
    }
    return fail;
}

// The last slot of a wave with 3 or 4 permutations per lane when it holds r <= 32 permutations (K = 136: 8; K = 150: 22;
// K = 200: 8): in the row loop its 64 - r idle lanes would cost every token a multiply all the same.  Instead the row loop
// leaves that slot out and, once the block's tile is in LDS, G = 64 / span lane groups -- each holding the same r
// permutations in its first r lanes (load_perms) -- take every G-th row of the tile each: 16 keys of the row from LDS
// (one address per group: broadcasts), their minimum tagged with the row, folded into the group's (smallest, second
// [, third]) record.  The records of the groups cover disjoint rows, so merging them across the groups (an xor butterfly
// over the group index) gives exactly the record the row loop would have produced: the proof that follows is the same.
template <int P, int STRIDE, int WPT, typename Rows>
__device__ __forceinline__ void share_last_slot(Rows &rec, const uint32_t *tile, const SievePerms<P> &sp, int rb, int lane) {
    const uint32_t span = sp.span, groups = (uint32_t)kWave / span;  // powers of two, groups in 2 .. 16
    const uint32_t mine = (uint32_t)lane / span;
    const uint32_t a_lo = sp.a_lo[P - 1];
    const uint64_t b8 = sp.b8[P - 1];
    rec = Rows();
#pragma unroll 1
    for (uint32_t r0 = 0; r0 < (uint32_t)rb; r0 += groups) {
        const uint32_t r = r0 + mine;                 // per lane group
        const uint32_t *rowp = tile + (r & 15u) * STRIDE;
        uint32_t m = kMaxHash;
#pragma unroll
        for (int c = 0; c < kRowTokens; c += 2) {  // (all 16 reads in flight: pinning four keys at a time as sieve_chunk does -- 69 instead of
            // 81 VGPRs -- measured 1-2 % slower: what this pass waits for is the LDS round trip, not a free wave slot)
            const uint32_t m0 = sieve_key(rowp[c * WPT], a_lo, b8);
            const uint32_t m1 = sieve_key(rowp[(c + 1) * WPT], a_lo, b8);
            m = umin3(m, m0, m1);
        }
        rec.add(r < (uint32_t)rb ? tag16(m, r) : kMaxHash);  // (2^32-1 changes nothing in the record)
    }
    for (uint32_t s = span; s < (uint32_t)kWave; s <<= 1) rec.merge_from_lane(lane ^ (int)s);
}

// Exact minima over the first nrows*16 tokens of [beg, ...) into res (min-combined); returns true
// in lanes whose proof failed (the caller redoes the range).  All arguments wave-uniform except the
// per-lane permutation registers.
template <int P, typename TokT, bool TIES = false, bool SHARE = false>
__device__ __forceinline__ bool sieve_range(const TokT MHX_CONST_AS *hv, const TokT *hv_vec, int64_t beg,
                                            int nrows, const Perms<P> &pm, const SievePerms<P> &sp,
                                            uint32_t *lds, int lane, uint32_t (&res)[P], int &nblocks, bool *tied = nullptr) {
    constexpr int N = Chunk<TokT>::N;
    constexpr int CPR = kRowTokens / N;  // chunks per row: 2 (uint64 tokens) or 1 (uint32)
    constexpr int STRIDE = StageLayout<TokT>::kStride, WPT = StageLayout<TokT>::kWordsPerTok;
    const TokT MHX_CONST_AS *p = hv + beg;
    const int nchunks = nrows * CPR;
    const auto chunk_ptr = [&](int idx) { return p + (int64_t)(idx < nchunks ? idx : nchunks - 1) * N; };
    Chunk<TokT> a, b;
    a.load(p);
    int ci = 0;  // chunk held by `a`
    bool fail = false;
    for (int r0 = 0; r0 < nrows; r0 += kBlockRows) {
        const int rb = min(kBlockRows, nrows - r0);
        const int64_t blk = beg + (int64_t)r0 * kRowTokens;
        // the block's tokens for the LDS tile: lane l carries 4 tokens (a quarter row); requested now,
        // needed after the row loop
        const int my_row = lane >> 2, my_part = lane & 3;
        uint4 st0 = {0, 0, 0, 0}, st1 = {0, 0, 0, 0};
        if (my_row < rb) {
            const uint4 *src = reinterpret_cast<const uint4 *>(hv_vec + blk + 4 * lane);
            st0 = src[0];
            if (WPT == 2) st1 = src[1];
        }
        typename std::conditional<TIES, Three, Two>::type rows[P];
        // Scalar loads return out of order, so the only wait is lgkmcnt(0): wait for the current
        // chunk (a use BEFORE the next prefetch is issued), THEN issue the prefetch, then hash.
        const auto row_loop = [&](auto slots) {
            constexpr int PS = decltype(slots)::value;
            if constexpr (CPR == 2) {
                // uint64 tokens: a row is the chunk pair (a, b); one row per iteration, nothing conditional
                for (int r = 0; r < rb; ++r) {
                    uint32_t row0[P];
                    a.arrived();
                    b.load(chunk_ptr(ci + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    sieve_chunk<P, TokT, true, PS>(a, sp, row0);
                    b.arrived();
                    a.load(chunk_ptr(ci + 2));
                    __builtin_amdgcn_sched_barrier(0);
                    sieve_chunk<P, TokT, false, PS>(b, sp, row0);
                    ci += 2;
#pragma unroll
                    for (int q = 0; q < PS; ++q) rows[q].add(tag16(row0[q], (uint32_t)r));
                }
            } else {
                // uint32 tokens: a row is one chunk; two rows per iteration keep the (a, b) ping-pong
                for (int r = 0; r < rb; r += 2) {
                    uint32_t row0[P], row1[P];
                    a.arrived();
                    b.load(chunk_ptr(ci + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    sieve_chunk<P, TokT, true, PS>(a, sp, row0);
                    b.arrived();
                    a.load(chunk_ptr(ci + 2));
                    __builtin_amdgcn_sched_barrier(0);
                    sieve_chunk<P, TokT, true, PS>(b, sp, row1);  // row r + 1
                    ci += 2;
                    const bool second = r + 1 < rb;  // wave-uniform
                    if (!second) ci -= 1;  // odd row count: chunk `b` was the clamped prefetch, not a row
#pragma unroll
                    for (int q = 0; q < PS; ++q) {
                        rows[q].add(tag16(row0[q], (uint32_t)r));
                        if (second) rows[q].add(tag16(row1[q], (uint32_t)(r + 1)));
                    }
                }
            }
        };
        constexpr bool kShare = SHARE && P >= 3;  // the last slot is left to share_last_slot (below)
        row_loop(std::integral_constant<int, (kShare ? P - 1 : P)>());
        // block complete: tile -> LDS (wave-private, so program order is enough: no barrier), then
        // rescan the best row of every permutation, prove, hash the one candidate
        {
            uint4 *dst = reinterpret_cast<uint4 *>(lds + my_row * STRIDE + my_part * (4 * WPT));
            dst[0] = st0;
            if (WPT == 2) dst[1] = st1;
        }
        if constexpr (kShare) share_last_slot<P, STRIDE, WPT>(rows[P - 1], lds, sp, rb, lane);
        if constexpr (TIES)
            fail |= finish_block_ties<P, STRIDE, WPT>(rows, lds, pm, sp, res, tied);
        else
            fail |= finish_block<P, STRIDE, WPT>(rows, lds, pm, sp, res);
        ++nblocks;
    }
    return fail;
}

// ---- the dedup launch: a flagged set with its repeated tokens dropped, then the sieve -----------------
// A set usually fails the sieve because a token occurs twice (two cells tie at the minimum).  Here each
// block of up to 256 tokens is copied into the wave's LDS tile and repeated tokens are found with a
// 1024-slot hash table in LDS (ds_min of the token index; a token whose slot holds a smaller index with the
// same value is a repeat).  A repeat is missed when an earlier DIFFERENT token sits in its slot (~10 % per
// pass), so the table is cleared and used again with another hash multiplier, three passes in all, which
// leaves ~0.1 % of the repeats (one table of 4 KB instead of two of 4 KB each: 6 instead of 3 waves per SIMD
// fit the LDS).  The survivors are compacted in place (ballot + mbcnt prefix) and the sieve runs over the
// compacted tile -- tokens now come from LDS as broadcast reads instead of scalar loads; the last, partial
// row takes part with the cells behind its tokens masked out.  What still fails the proof (a missed repeat,
// keys that are genuinely within 32) is left to the pairwise launch.
constexpr int kDedupSlots = 1024;
constexpr int kDedupPasses = 3;
constexpr int kDedupWordsPerWave = kStageWordsPerWave + kDedupSlots;

// lane l's four tokens of the block that starts at blk (zero where the block has ended)
template <typename TokT>
__device__ __forceinline__ void load_quad(const TokT *hv_vec, int64_t blk, int64_t end, int lane, uint64_t (&t)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = blk + 4 * lane + i < end ? (uint64_t)hv_vec[blk + 4 * lane + i] : 0;
}

// `first`: the quad of the first block, loaded by the caller (fetching the NEXT flagged set's quad a set ahead was
// measured: 8 VGPRs live across the whole loop, 92 -> 100, and 1-2 % slower on every corpus)
template <int P, typename TokT>
__device__ __forceinline__ bool dedup_sieve_range(const TokT *hv_vec, int64_t beg, int64_t end, const Perms<P> &pm,
                                                  const SievePerms<P> &sp, uint32_t *lds, int lane,
                                                  const uint64_t (&first)[4], uint32_t (&res)[P]) {
    constexpr int STRIDE = 36;  // the tile always holds 64-bit tokens here
    uint32_t *tile = lds;
    uint32_t *table = lds + kStageWordsPerWave;  // [kDedupSlots] token indices
    bool fail = false;
    for (int64_t blk = beg; blk < end; blk += kBlockRows * kRowTokens) {
        const int nb = (int)min((int64_t)(kBlockRows * kRowTokens), end - blk);
        // lane l owns tokens 4l .. 4l+3 of the block
        uint64_t t[4];
        bool keep[4];
        if (blk == beg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = first[i];
        } else {
            load_quad<TokT>(hv_vec, blk, end, lane, t);
        }
        uint32_t mix[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t idx = 4 * lane + i;
            keep[i] = (int)idx < nb;
            mix[i] = (uint32_t)t[i] ^ ((uint32_t)(t[i] >> 32) * 0x9E3779B1u);
            uint32_t *cell = tile + (idx >> 4) * STRIDE + (idx & 15u) * 2;  // raw tile: token idx at (row idx/16, column idx%16)
            cell[0] = (uint32_t)t[i];
            cell[1] = (uint32_t)(t[i] >> 32);
        }
        for (int pass = 0; pass < kDedupPasses; ++pass) {
            const uint32_t mult = pass == 0 ? 0x9E3779B1u : pass == 1 ? 0x85EBCA77u : 0xC2B2AE3Du;
#pragma unroll
            for (int i = 0; i < kDedupSlots / kWave; i += 4)
                *reinterpret_cast<uint4 *>(table + (kDedupSlots / kWave) * lane + i) = uint4{~0u, ~0u, ~0u, ~0u};
            // Straight-line code, three LDS round trips per pass for all four tokens of the lane together (with a
            // branch per token the dependent reads queued up one behind the other: ~10 round trips per pass).  A
            // token that is already out takes part with the index 2^32-1, which never wins a slot.
            uint32_t slot[4], owner[4], c0[4], c1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                slot[i] = (mix[i] * mult) >> 22;
                atomicMin(table + slot[i], keep[i] ? (uint32_t)(4 * lane + i) : ~0u);
            }
            // LDS operations of one wave complete in order: every lane's ds_min is done before the reads below
#pragma unroll
            for (int i = 0; i < 4; ++i) owner[i] = table[slot[i]];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t o = owner[i] & 255u;  // a slot nobody took holds 2^32-1: any cell will do, the test below fails
                const uint32_t *cell = tile + (o >> 4) * STRIDE + (o & 15u) * 2;
                c0[i] = cell[0];
                c1[i] = cell[1];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool repeat = owner[i] < (uint32_t)(4 * lane + i) && c0[i] == (uint32_t)t[i] && c1[i] == (uint32_t)(t[i] >> 32);
                keep[i] = keep[i] && !repeat;
            }
        }
        // compact in place (every read of the raw tile is done: the tile is private to the wave)
        uint32_t below = 0;  // kept tokens of lower lanes
        int kept = 0;        // kept tokens of the block (wave-uniform)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long mask = __ballot(keep[i]);
            below += __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            kept += __popcll(mask);
        }
        uint32_t pos = below;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (keep[i]) {
                uint32_t *cell = tile + (pos >> 4) * STRIDE + (pos & 15u) * 2;
                cell[0] = (uint32_t)t[i];
                cell[1] = (uint32_t)(t[i] >> 32);
                ++pos;
            }
        }
        // sieve over the rows of the compacted tile: wave-uniform LDS addresses = broadcast reads
        const int nrows = kept >> 4, rest = kept & 15;
        Two rows[P];
        for (int r = 0; r < nrows; ++r) {
            const uint4 *rowq = reinterpret_cast<const uint4 *>(tile + r * STRIDE);
            uint32_t row[P];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                uint4 two[4];
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) two[c2] = rowq[4 * half + c2];  // tokens 2*c2 and 2*c2+1: low words .x and .z
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) {
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const uint32_t k0 = sieve_key(two[c2].x, sp.a_lo[q], sp.b8[q]);
                        const uint32_t k1 = sieve_key(two[c2].z, sp.a_lo[q], sp.b8[q]);
                        row[q] = (half == 0 && c2 == 0) ? min(k0, k1) : umin3(row[q], k0, k1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // eight tokens in registers at a time
            }
#pragma unroll
            for (int q = 0; q < P; ++q) rows[q].add(tag16(row[q], (uint32_t)r));
        }
        if (rest > 0) {  // the partial last row (nrows <= 15 here: a full tile has no rest)
            uint32_t row[P];
#pragma unroll
            for (int q = 0; q < P; ++q) row[q] = kMaxHash;
            for (int c = 0; c < rest; ++c) {
                const uint32_t lo = tile[nrows * STRIDE + 2 * c];
#pragma unroll
                for (int q = 0; q < P; ++q) row[q] = min(row[q], sieve_key(lo, sp.a_lo[q], sp.b8[q]));
            }
#pragma unroll
            for (int q = 0; q < P; ++q) rows[q].add(tag16(row[q], (uint32_t)nrows));
        }
        if (kept > 0) fail |= finish_block<P, STRIDE, 2, true>(rows, tile, pm, sp, res, (uint32_t)nrows, (uint32_t)rest);
    }
    return fail;
}

template <int P>
struct Minima {
    uint32_t v[P];
};

// Full evaluation of the tokens [beg,end): fast fold, then the exact fold iff some lane's minimum
// lands in the ambiguous zone (or straight away with exact_only).  In the default path it runs for
// a few sets per ten thousand (failed sieve proofs), so it is written to cost the hot path
// nothing: it reloads the permutations of its lane itself and its address arithmetic is fenced
// off from loop-invariant hoisting (a real call would add the callee's registers to the kernel's).
template <int P, typename TokT>
__device__ __forceinline__ Minima<P> full_minima(const TokT *hv_vec, int64_t beg, int64_t end, const uint64_t *a,
                                                 const uint64_t *b, int num_perm, int kbase, bool exact_only,
                                                 unsigned long long *stats) {
    // Launder the range: nothing computed from it below can be hoisted above this point.
    asm volatile("" : "+s"(beg), "+s"(end));
    const int lane = threadIdx.x & (kWave - 1);
    const TokT MHX_CONST_AS *hv = as_const(hv_vec);
    Perms<P> pm, pm_biased;
    bool active[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int k = kbase + p * kWave + lane;
        active[p] = k < num_perm;
        const uint64_t av = active[p] ? a[k] : 0;
        pm.a_lo[p] = pm_biased.a_lo[p] = (uint32_t)av;
        pm.a_hi[p] = pm_biased.a_hi[p] = (uint32_t)(av >> 32);
        pm.b[p] = active[p] ? b[k] : 0;
        pm_biased.b[p] = pm.b[p] + 1;  // wraps mod 2^64 like everything else
    }
    Minima<P> res;
    bool redo = exact_only;
    if (!exact_only) {
        uint32_t acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = kMaxHash;
        hash_range<P, false, TokT>(hv, beg, end, pm_biased, acc);
        bool suspicious = false;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            suspicious |= active[p] && acc[p] <= 7u;
            res.v[p] = acc[p] - 1u;
        }
        redo = __any(suspicious);
    }
    if (redo) {
        if (!exact_only && stats && lane == 0) atomicAdd(stats + 1, 1ull);
#pragma unroll
        for (int p = 0; p < P; ++p) res.v[p] = kMaxHash;
        hash_range<P, true, TokT>(hv, beg, end, pm, res.v);
    }
    return res;
}

// Sieve over the full 16-token rows of [beg,end) plus fast fold for the ragged tail.  Returns true
// (wave-uniform) when a proof failed or a tail minimum is ambiguous: the caller then redoes the
// range with full_minima.
template <int P, typename TokT, bool TAIL = true, bool TIES = false, bool SHARE = false>
__device__ __forceinline__ bool sieve_minima(const TokT MHX_CONST_AS *hv, const TokT *hv_vec, int64_t beg,
                                             int64_t end, const Perms<P> &pm, const Perms<P> &pm_biased,
                                             const SievePerms<P> &sp, unsigned long long *stats, int lane,
                                             uint32_t *lds, uint32_t (&res)[P], bool *tied = nullptr) {
    const int64_t n = end - beg;
    const int nrows = (int)min(n / kRowTokens, (int64_t)(1 << 27));
#pragma unroll
    for (int p = 0; p < P; ++p) res[p] = kMaxHash;
    bool bad = false;
    if (nrows > 0) {
        int nblocks = 0;
        bad = sieve_range<P, TokT, TIES, SHARE>(hv, hv_vec, beg, nrows, pm, sp, lds, lane, res, nblocks, tied);
        if (stats && lane == 0) atomicAdd(stats + 2, (unsigned long long)nblocks);
    }
    const int64_t tail = beg + (int64_t)nrows * kRowTokens;
    if (TAIL && tail < end) {  // TAIL == false: the caller guarantees whole rows (no fast-fold code, 48 instead of 73 VGPRs)
        uint32_t acc[P];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p] = kMaxHash;
        hash_range<P, false, TokT>(hv, tail, end, pm_biased, acc);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            bad |= sp.active[p] && acc[p] <= 7u;
            res[p] = min(res[p], acc[p] - 1u);
        }
    }
    return __any(bad);
}

// Per-wave back-off for corpora whose sets defeat the sieve (repeated tokens inside a set): after
// two failed proofs in a row the wave skips the sieve for `gap` ranges and sends them straight to
// the full evaluation; the gap doubles with every further failure (16 .. 64) and resets on a
// success.  A lone failure (4 in 10^4 sets of distinct tokens) changes nothing; corpora full of
// repeats pay the sieve on 1 range in 64 instead of on all.
struct SieveBackoff {
    int skip = 0;
    int gap = 16;
    int streak = 0;
    __device__ __forceinline__ void failed() {
        if (++streak >= 2) {
            skip = gap;
            gap = min(2 * gap, 64);
        }
    }
    __device__ __forceinline__ void succeeded() {
        streak = 0;
        gap = 16;
    }
};

// The second launch tries the tie-tolerant sieve first and the dedup pass where that fails.  Trying pays while fewer
// than about one set in five fails (1 360 instructions tried, 1 730 for the dedup pass): every failure adds 4 to a
// score, every success takes 1 off; at 8 the wave stops trying for 32 sets, then looks again.
struct TiesBackoff {
    int skip = 0;
    int score = 0;
    __device__ __forceinline__ void failed() {
        score += 4;
        if (score >= 8) {
            skip = 32;
            score = 4;
        }
    }
    __device__ __forceinline__ void succeeded() { score = max(score - 1, 0); }
};

// min over tokens [beg,end) of the exact fold, for the P permutations of this lane.
//   path 0: sieve; a failed proof sends the whole range to the full evaluation
//   path 2: full evaluation (fast fold, exact redo)     path 1: exact fold for every pair
template <int P, typename TokT>
__device__ __forceinline__ void set_minima(const BulkArgs &args, const TokT MHX_CONST_AS *hv, const TokT *hv_vec,
                                           int64_t beg, int64_t end, const Perms<P> &pm,
                                           const Perms<P> &pm_biased, const SievePerms<P> &sp, int kbase,
                                           int lane, uint32_t *lds, SieveBackoff &bo, uint32_t (&res)[P]) {
    bool full = args.path != 0;
    if (!full) {
        if (bo.skip > 0) {
            --bo.skip;
            full = true;
        } else {
            full = sieve_minima<P, TokT>(hv, hv_vec, beg, end, pm, pm_biased, sp, args.stats, lane, lds, res);
            if (full) {
                if (args.stats && lane == 0) atomicAdd(args.stats, 1ull);
                bo.failed();
            } else {
                bo.succeeded();
            }
        }
    }
    if (full) {
        const Minima<P> m = full_minima<P, TokT>(hv_vec, beg, end, args.a, args.b, args.num_perm, kbase,
                                                 args.path == 1, args.stats);
#pragma unroll
        for (int p = 0; p < P; ++p) res[p] = m.v[p];
    }
}

template <int P, bool SHARE = false>
__device__ __forceinline__ void load_perms(const BulkArgs &args, int kbase, int lane, Perms<P> &pm,
                                           Perms<P> &pm_biased, SievePerms<P> &sp, int (&kidx)[P]) {
    // SHARE (the launcher's choice: one pass over the permutations, r <= 32 of them left for the last slot of 3 or 4):
    // 64/span lane groups hold the same r (share_last_slot); only the first group's lanes own them (kidx, active: they
    // store, their proofs count)
    uint32_t span = kWave;
    if constexpr (SHARE && P >= 3) {
        const int r = args.num_perm - (kbase + (P - 1) * kWave);
        span = r > 16 ? 32u : r > 8 ? 16u : r > 4 ? 8u : 4u;
    }
    sp.span = span;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool shared = SHARE && P >= 3 && p == P - 1;
        const int k = kbase + p * kWave + (shared ? lane & (int)(span - 1) : lane);
        const bool holds = k < args.num_perm;
        kidx[p] = holds && !(shared && lane >= (int)span) ? k : -1;
        const uint64_t a = holds ? args.a[k] : 0;
        pm.a_lo[p] = (uint32_t)a;
        pm.a_hi[p] = (uint32_t)(a >> 32);
        pm.b[p] = holds ? args.b[k] : 0;
        pm_biased.a_lo[p] = pm.a_lo[p];
        pm_biased.a_hi[p] = pm.a_hi[p];
        pm_biased.b[p] = pm.b[p] + 1;  // wraps mod 2^64 like everything else
        sp.a_lo[p] = pm.a_lo[p];
        sp.b8[p] = pm.b[p] + 8;
        sp.active[p] = kidx[p] >= 0;
    }
}

// ---- kernel A: one wave per set, in two launches ------------------------------------------------
// grid.x strides over sets; a wave walks the permutations of its set in chunks of 64*P (the
// set's tokens stay in the scalar cache / L2 between chunks), so a [K] row is written by one wave.
//   MODE_SIEVE  the sieve and nothing else.  A set whose proof fails (or that the wave's back-off
//               skips) is appended to the redo list instead of being evaluated in full here: keeping
//               the full evaluation out of this kernel is worth 8 % (60 instead of 122 VGPRs, no
//               hoisted address arithmetic for paths that run four times in ten thousand sets).
//   MODE_FULL   the full evaluation (fast fold + exact redo, or the exact fold with path 1) for the
//               listed sets -- or for every set when the sieve is switched off.
//   MODE_DEDUP  between the two: the sets the sieve flagged get the dedup sieve (repeated tokens dropped in LDS, sieve
//               over what is left) and NOTHING else -- a set it cannot settle either is re-flagged for MODE_FULL.  Without
//               the pair-by-pair code this launch needs far fewer registers than MODE_FULL, and a corpus full of
//               repeated tokens is settled here at close to the sieve's own rate.
//   MODE_SIEVE_TIES  (round 4) MODE_SIEVE with the proof that tolerates one tie (finish_block_ties), as a kernel of its own:
//               on a corpus whose sets repeat tokens 97 % of the sets fail the one-candidate proof, the sieve launch is
//               wasted on them and the tie-tolerant proof then runs in the second launch at its occupancy and with its flag
//               scan.  Both first launches are always enqueued; which one works is decided ON THE DEVICE from what the
//               previous call on the context learned (BulkArgs::mode_word, written by the last launch of every call): the
//               other returns at once (~10 us).  No host read-back, nothing added to the hot kernel's per-set code.
enum { MODE_SIEVE = 0, MODE_FULL = 1, MODE_DEDUP = 2, MODE_SIEVE_TIES = 3 };
enum { SHAPE_GENERAL = 0, SHAPE_PLAIN = 1, SHAPE_PLAIN_FIXED = 2, SHAPE_PLAIN_FIXED_ROWS = 3 };

template <int P, typename TokT, typename OutT, int MODE, int SHAPE, bool SHARE = false>
__global__ __launch_bounds__(256) void minhash_bulk_kernel(const BulkArgs args_in) {
    // SHAPE_PLAIN: no initial state, no counters, no aliasing; SHAPE_PLAIN_FIXED: and fixed-length sets;
    // SHAPE_PLAIN_FIXED_ROWS: and the length is a multiple of 16, so the fast-fold code for a ragged tail is
    // not compiled in at all.  The fields become compile-time constants, so their SGPRs and address arithmetic
    // leave the per-set path (spilled SGPRs 49 -> 32 -> 4, VGPRs 78 -> 73 -> 53): 3 % of the headline launch.
    BulkArgs args = args_in;
    constexpr bool kSieve = MODE == MODE_SIEVE || MODE == MODE_SIEVE_TIES;
    if (kSieve && args.mode_word) {  // (wave-uniform) the first launch that the corpus calls for works, the other one leaves
        const unsigned int want_ties = *as_const(args.mode_word);
        if ((want_ties == 1u) != (MODE == MODE_SIEVE_TIES)) return;
    }
    if (MODE == MODE_FULL && args.mode_word && args.sieve_hint && blockIdx.x == 0 && threadIdx.x == 0) {
        // the last launch of the call: what the first launch saw decides the next call's first launch (a quarter of the
        // sets defeating the one-candidate proof is where the tie-tolerant one, 1.18x the instructions, starts to pay)
        // 0: the one-candidate proof first; 1: the tie-tolerant proof first; 2: the one-candidate proof first because even the
        // tie-tolerant one left a fifth of the sets to the dedup pass last time it ran first (10 % repeated tokens: 2.16 ms per
        // 500k sets that way round against 2.0) -- until the corpus changes
        const unsigned int t = __builtin_nontemporal_load(args.sieve_hint), f = __builtin_nontemporal_load(args.sieve_hint + 1);
        const unsigned int g = __builtin_nontemporal_load(args.sieve_hint + 2), m = *args.mode_word;
        // (what the second launch saw when IT tried the tie-tolerant proof, words 3 and 5: tried / left to its dedup pass)
        const unsigned int tt = __builtin_nontemporal_load(args.sieve_hint + 3), tf = __builtin_nontemporal_load(args.sieve_hint + 5);
        if (t >= 64u) {
            const bool defeated = 4u * f > t;
            const bool heavy = m == 1u ? 5u * g > t : (tt >= 64u ? 5u * tf > tt : m == 2u);
            *args.mode_word = !defeated ? 0u : heavy ? 2u : 1u;
        }
    }
    if (SHAPE != SHAPE_GENERAL) {
        args.init = nullptr;
        args.init_stride = 0;
        args.stats = nullptr;
        args.alias_mask = -1;
    }
    if (SHAPE == SHAPE_PLAIN_FIXED || SHAPE == SHAPE_PLAIN_FIXED_ROWS) args.offsets = nullptr;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    const int kchunks = (args.num_perm + kWave * P - 1) / (kWave * P);
    // LDS per wave: the token tile of the rescan; the full launch adds the hash tables of its dedup sieve
    constexpr int kLdsPerWave = MODE == MODE_DEDUP ? kDedupWordsPerWave : kStageWordsPerWave;
    __shared__ __attribute__((aligned(16))) uint32_t stage[4 * kLdsPerWave];
    uint32_t *lds = stage + wave * kLdsPerWave;
    Perms<P> pm, pm_biased;
    SievePerms<P> sp;
    int kidx[P];
    if (kSieve || MODE == MODE_DEDUP) load_perms<P, SHARE>(args, 0, lane, pm, pm_biased, sp, kidx);

    const TokT MHX_CONST_AS *hv = as_const(static_cast<const TokT *>(args.hv));
    const TokT *hv_vec = static_cast<const TokT *>(args.hv);
    const int64_t MHX_CONST_AS *offsets = as_const(args.offsets);
    OutT *__restrict__ out = static_cast<OutT *>(args.out);
    const int64_t stride = (int64_t)gridDim.x * waves_per_block;
    // MODE_FULL after a sieve launch walks the flags 64 sets at a time: one byte load per lane, a
    // ballot, then the flagged sets one by one (a flag array, not an appended list: hundreds of
    // thousands of atomics on one counter serialise -- measured 8 ms for 500k failed sets)
    const bool flagged_only = !kSieve && args.redo != nullptr;
    // (the four waves of a workgroup share one group of 64 flags and take every fourth flagged set of it:
    // one wave per 64 sets left 2.5 rounds of 64-set waves when everything was flagged)
    // the pairwise launch takes its sets from the list when all of them fit in it (no flag scan: 41 -> ~15 us per step
    // on a clean corpus, where it has a handful of sets among a million flags)
    unsigned int listed_n = 0;
    bool listed = false;
    if (MODE == MODE_FULL && flagged_only && args.pair_list) {
        listed_n = *as_const(args.pair_count);
        listed = listed_n <= kPairListCap;
    }
    const int64_t n_items = listed ? (int64_t)listed_n : flagged_only ? (args.n_sets + kWave - 1) / kWave * waves_per_block : args.n_sets;
    SieveBackoff backoff;
    TiesBackoff ties;
    int tried = 0, failed = 0, left = 0;  // the first launch: this wave's contribution to args.sieve_hint (left: MODE_SIEVE_TIES, sets it could not settle)
    if (MODE == MODE_SIEVE && args.sieve_hint) {  // (the tie-tolerant first launch never skips: what it cannot settle is rare)
        // a launch over a corpus whose sets defeat the sieve (repeated tokens) should not find that out wave by wave
        // (a wave sees only a handful of sets): waves publish their counts when they leave, and a wave that starts
        // after most proofs have failed begins in the skipping state
        const unsigned int t = __builtin_nontemporal_load(args.sieve_hint), f = __builtin_nontemporal_load(args.sieve_hint + 1);
        if (t >= 64u && 2u * f > t) {
            backoff.skip = 63;
            backoff.gap = 64;
            backoff.streak = 2;
        }
    }
    // The pairwise launch over a list: a lone wave per set would pay one memory round trip per 64-byte scalar load, one
    // after the other (~20 us for 256 tokens, and a clean corpus leaves this launch a handful of sets and nothing to
    // hide that behind).  So the four waves of a workgroup take a quarter of the set's tokens each and combine their
    // minima through LDS.
    if (MODE == MODE_FULL && listed) {
        uint32_t *comb = stage;  // [4 waves][P][64 lanes]
        for (int64_t it = blockIdx.x; it < (int64_t)listed_n; it += gridDim.x) {
            const int64_t set = (int64_t)as_const(args.pair_list)[it];
            const int64_t beg = args.offsets ? offsets[set] : set * args.fixed_len;
            const int64_t end = args.offsets ? offsets[set + 1] : beg + args.fixed_len;
            const int64_t share = (end - beg + 31) / 32 * 8;  // whole 8-token chunks
            const int64_t wbeg = min(end, beg + wave * share), wend = min(end, wbeg + share);
            for (int kc = 0; kc < kchunks; ++kc) {
                uint32_t mine[P];
#pragma unroll
                for (int p = 0; p < P; ++p) mine[p] = kMaxHash;
                if (wend > wbeg) {
                    const Minima<P> m = full_minima<P, TokT>(hv_vec, wbeg, wend, args.a, args.b, args.num_perm, kc * (kWave * P),
                                                             args.path == 1, args.stats);
#pragma unroll
                    for (int p = 0; p < P; ++p) mine[p] = m.v[p];
                }
#pragma unroll
                for (int p = 0; p < P; ++p) comb[(wave * P + p) * kWave + lane] = mine[p];
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        const int k = kc * (kWave * P) + p * kWave + lane;
                        if (k >= args.num_perm) continue;
                        uint32_t r = mine[p];
#pragma unroll
                        for (int w = 1; w < 4; ++w) r = min(r, comb[(w * P + p) * kWave + lane]);
                        uint64_t v = r;
                        if (args.init) {
                            const uint64_t iv = args.init[set * args.init_stride + k];
                            v = end > beg ? (uint64_t)min((iv >> 32) ? kMaxHash : (uint32_t)iv, r) : iv;
                        }
                        if (sizeof(OutT) == 4) v = v > kMaxHash ? kMaxHash : v;
                        out[set * args.num_perm + k] = (OutT)v;
                    }
                }
                __syncthreads();
            }
        }
        return;
    }
    // flagged launches: the (flag group, wave) items go round the waves of a grid that is exactly as large as what
    // is resident at once (one atomic counter handing them out one by one serialises: 62 500 items at ~90 dequeues per
    // microsecond cost 0.7 ms per launch; a larger static grid leaves its last round of workgroups running alone)
    for (int64_t item0 = (int64_t)blockIdx.x * waves_per_block + wave; item0 < n_items; item0 += stride) {
      const int64_t item = flagged_only ? item0 / waves_per_block : item0;  // 64-flag group
      unsigned long long todo = 1;  // sets of this item still to do (bit i = set 64*item + i when flagged_only)
      if (flagged_only && !listed) {
          const int64_t cand = item * kWave + lane;
          const bool mine = (lane % waves_per_block) == (int)(item0 % waves_per_block);  // every fourth flagged set of the group
          todo = __ballot(mine && cand < args.n_sets && args.redo[cand] == (uint8_t)args.redo_match);
      }
      while (todo) {
        const int bit = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int64_t set = listed ? (int64_t)as_const(args.pair_list)[item0] : flagged_only ? item * kWave + bit : item;
        int64_t beg, end;
        if (args.offsets) {
            beg = offsets[set];
            end = offsets[set + 1];
        } else {
            beg = (args.alias_mask >= 0 ? (set & args.alias_mask) : set) * args.fixed_len;
            end = beg + args.fixed_len;
        }
        // Warm L2 / Infinity Cache with the tokens of the set this wave hashes NEXT: one vector load
        // per wave per set, lane l touching byte 128*l of that set (up to 8 KiB).  The scalar loads
        // of the next iteration then hit on-chip instead of paying an HBM round trip each.
        uint32_t warm = 0;
        {
            const int64_t nset = listed ? args.n_sets : flagged_only ? (todo ? item * kWave + __builtin_ctzll(todo) : args.n_sets) : item + stride;
            // (a wave that is skipping the sieve -- the corpus defeats it -- does not read the next set either: the
            // warm-up loads of a skipping launch alone kept it at 0.18 ms per 500k sets, one pass over the corpus)
            const bool skipping_on = MODE == MODE_SIEVE && backoff.skip > 1;
            if (args.prefetch && nset < args.n_sets && !skipping_on) {
                const int64_t nbeg = args.offsets ? offsets[nset] : nset * args.fixed_len;
                const int64_t nend = args.offsets ? offsets[nset + 1] : nbeg + args.fixed_len;
                const int64_t off = (int64_t)lane * 128;
                if (off < (nend - nbeg) * (int64_t)sizeof(TokT))
                    warm = *reinterpret_cast<const volatile uint32_t *>(reinterpret_cast<const char *>(args.hv) +
                                                                         nbeg * (int64_t)sizeof(TokT) + off);
            }
        }
        bool defer = false;  // MODE_SIEVE: leave this set to the MODE_FULL launch
        if (MODE == MODE_SIEVE && end > beg && backoff.skip > 0) {
            --backoff.skip;
            defer = true;
        }
        for (int kc = 0; kc < kchunks && !defer; ++kc) {
            uint32_t res[P];
            if (kSieve) {
                if (kchunks > 1) load_perms<P, SHARE>(args, kc * (kWave * P), lane, pm, pm_biased, sp, kidx);
                if (end > beg) {
                    bool tied = false;
                    defer = sieve_minima<P, TokT, SHAPE != SHAPE_PLAIN_FIXED_ROWS, MODE == MODE_SIEVE_TIES, SHARE>(hv, hv_vec, beg, end, pm, pm_biased, sp, args.stats, lane, lds, res,
                                                                                                          MODE == MODE_SIEVE_TIES ? &tied : nullptr);
                    if (kc == 0) ++tried;
                    if (MODE == MODE_SIEVE_TIES) {  // "failed" = the one-candidate proof would have: the corpus still calls for this kernel
                        if (defer || __any(tied)) ++failed;
                        if (defer) {
                            ++left;
                            break;
                        }
                    } else {
                        if (defer) {
                            ++failed;
                            backoff.failed();
                            break;
                        }
                        backoff.succeeded();
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int k = kc * (kWave * P) + p * kWave + lane;
                    kidx[p] = k < args.num_perm ? k : -1;
                }
                if (end > beg) {
                    if (MODE == MODE_DEDUP) {
                        // a flagged set gets the dedup sieve (repeated tokens are what usually broke the proof); what
                        // that cannot prove either is left to the pairwise launch
                        if (kchunks > 1) load_perms<P>(args, kc * (kWave * P), lane, pm, pm_biased, sp, kidx);
                        // the sieve again with the proof that tolerates one tie (a token occurring twice, two close keys)
                        bool open = true;
                        if (!args.ties || (args.mode_word && *as_const(args.mode_word) == 1u)) {  // (the tie-tolerant proof has been tried: by the first launch)
                        } else if (ties.skip > 0) {
                            if (kc == kchunks - 1) --ties.skip;
                        } else {
                            open = sieve_minima<P, TokT, true, true>(hv, hv_vec, beg, end, pm, pm_biased, sp, nullptr, lane, lds, res);
                            if (kc == 0) ++tried, failed += open ? 1 : 0;
                            if (open) ties.failed(); else ties.succeeded();
                        }
                        if (open) {
#pragma unroll
                        for (int p = 0; p < P; ++p) res[p] = kMaxHash;
                        uint64_t cur[4];
                        load_quad<TokT>(hv_vec, beg, end, lane, cur);
                        defer = __any(dedup_sieve_range<P, TokT>(hv_vec, beg, end, pm, sp, lds, lane, cur, res));
                        if (defer) {
                            if (args.stats && lane == 0) atomicAdd(args.stats + 3, 1ull);
                            break;
                        }
                        }
                    } else {
                        const Minima<P> m = full_minima<P, TokT>(hv_vec, beg, end, args.a, args.b, args.num_perm,
                                                                 kc * (kWave * P), args.path == 1, args.stats);
#pragma unroll
                        for (int p = 0; p < P; ++p) res[p] = m.v[p];
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (kidx[p] < 0) continue;
                uint64_t v;
                if (args.init) {
                    const uint64_t iv = args.init[set * args.init_stride + kidx[p]];
                    if (end > beg) {
                        const uint32_t ic = (iv >> 32) ? kMaxHash : (uint32_t)iv;
                        v = min(ic, res[p]);
                    } else {
                        v = iv;  // empty set: state untouched (minhash.py:265-266)
                    }
                } else {
                    v = end > beg ? res[p] : kMaxHash;
                }
                if (sizeof(OutT) == 4) v = v > kMaxHash ? kMaxHash : v;
                out[set * args.num_perm + kidx[p]] = (OutT)v;
            }
        }
        if (kSieve && lane == 0) {
            args.redo[set] = defer ? 1 : 0;
            if (defer && args.stats) atomicAdd(args.stats, 1ull);
        }
        if (MODE == MODE_DEDUP && defer && lane == 0) {
            args.redo[set] = 2;
            if (args.pair_list) {  // (rare: a few sets in 10^5 of a clean corpus, ~1 % of a corpus full of repeats)
                const unsigned int slot = atomicAdd(args.pair_count, 1u);
                if (slot < kPairListCap) args.pair_list[slot] = (unsigned int)set;
            }
        }
        asm volatile("" ::"v"(warm));  // the warm-up load retires here, a whole set later
      }
    }
    // (a sample of the waves publishes: atomics of all 65 536 waves on one word would serialise for over a millisecond)
    if (kSieve && args.sieve_hint && lane == 0 && wave == 0 && (blockIdx.x & 15u) == 0 && tried > 0) {
        atomicAdd(args.sieve_hint, (unsigned int)tried);
        if (failed) atomicAdd(args.sieve_hint + 1, (unsigned int)failed);
        if (left) atomicAdd(args.sieve_hint + 2, (unsigned int)left);
    }
    if (MODE == MODE_DEDUP && args.sieve_hint && args.mode_word && lane == 0 && tried > 0) {
        atomicAdd(args.sieve_hint + 3, (unsigned int)tried);
        if (failed) atomicAdd(args.sieve_hint + 5, (unsigned int)failed);
    }
}

// ---- kernel C: several sets per wave for short signatures (num_perm <= 32) ----------------------------------
// With one set per wave, lanes >= num_perm idle (K = 16: three quarters of the machine).  Here a wave takes
// G = 64 / KP sets at once (KP = 8, 16 or 32 lanes per set, one permutation per lane).  The sets' tokens then differ
// between the lane groups, so they cannot be scalar operands: every block of up to 256 tokens of each of the G sets
// is copied, coalesced, into the group's own LDS tile (the tile of the rescan, now also the source of the hot loop:
// a lane group reads the same address, a broadcast), and the sieve runs per lane exactly as in kernel A -- row minima
// of 16 keys, tagged top-two fold, rescan of the best row, one exact candidate; the partial last row takes part with
// its stale cells masked.  A set whose proof fails in any of its lanes is flagged for the dedup / pairwise launches.
template <int KP, int P, typename TokT, typename OutT>
__global__ __launch_bounds__(256) void minhash_packed_kernel(const BulkArgs args) {
    constexpr int G = kWave / KP;
    constexpr int STRIDE = 36;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane / KP, k = lane % KP;
    __shared__ __attribute__((aligned(16))) uint32_t stage[4 * G * kStageWordsPerWave];
    uint32_t *tiles = stage + wave * (G * kStageWordsPerWave);
    uint32_t *my_tile = tiles + g * kStageWordsPerWave;
    // lane k of a group holds permutations k, k + KP, ... (P of them): a token read from the tile serves all P
    Perms<P> pm;
    SievePerms<P> sp;
    bool any_active = false;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const bool active = k + q * KP < args.num_perm;
        const uint64_t a = active ? args.a[k + q * KP] : 0, b = active ? args.b[k + q * KP] : 0;
        pm.a_lo[q] = sp.a_lo[q] = (uint32_t)a;
        pm.a_hi[q] = (uint32_t)(a >> 32);
        pm.b[q] = b;
        sp.b8[q] = b + 8;
        sp.active[q] = active;
        any_active |= active;
    }
    const TokT *hv_vec = static_cast<const TokT *>(args.hv);
    OutT *__restrict__ out = static_cast<OutT *>(args.out);
    const unsigned long long group_lanes = (KP == 64 ? ~0ull : ((1ull << KP) - 1ull)) << (g * KP);
    const int64_t n_items = (args.n_sets + G - 1) / G;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < n_items; item += (int64_t)gridDim.x * 4) {
        const int64_t set = item * G + g;  // per lane group
        const bool has = set < args.n_sets;
        int64_t beg = 0, end = 0;
        if (has) {
            beg = args.offsets ? args.offsets[set] : set * args.fixed_len;
            end = args.offsets ? args.offsets[set + 1] : beg + args.fixed_len;
        }
        uint32_t res[P];
#pragma unroll
        for (int q = 0; q < P; ++q) res[q] = kMaxHash;
        bool fail = false;
        for (int64_t boff = 0; __any(beg + boff < end); boff += kBlockRows * kRowTokens) {
            const int64_t blk = beg + boff;
            const int nb = (int)max((int64_t)0, min((int64_t)(kBlockRows * kRowTokens), end - blk));  // my set's tokens in this block
            // the G tiles, filled by the whole wave one after the other: lane l carries tokens 4l .. 4l+3
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int jnb = __builtin_amdgcn_readlane(nb, j * KP);
                if (jnb == 0) continue;  // wave-uniform
                const uint32_t blo = __builtin_amdgcn_readlane((int)(uint32_t)blk, j * KP);
                const uint32_t bhi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)blk >> 32), j * KP);
                const int64_t jblk = (int64_t)(((uint64_t)bhi << 32) | blo);
                uint64_t t[4] = {0, 0, 0, 0};
                if (4 * lane + 3 < jnb) {  // the usual case: all four in range, one or two wide loads
                    if constexpr (sizeof(TokT) == 8) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(hv_vec + jblk + 4 * lane);
                        const uint4 lo = src[0], hi = src[1];
                        t[0] = ((uint64_t)lo.y << 32) | lo.x;
                        t[1] = ((uint64_t)lo.w << 32) | lo.z;
                        t[2] = ((uint64_t)hi.y << 32) | hi.x;
                        t[3] = ((uint64_t)hi.w << 32) | hi.z;
                    } else {
                        const uint4 v = *reinterpret_cast<const uint4 *>(hv_vec + jblk + 4 * lane);
                        t[0] = v.x;
                        t[1] = v.y;
                        t[2] = v.z;
                        t[3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) t[i] = 4 * lane + i < jnb ? (uint64_t)hv_vec[jblk + 4 * lane + i] : 0;
                }
                uint4 *dst = reinterpret_cast<uint4 *>(tiles + j * kStageWordsPerWave + (lane >> 2) * STRIDE + (lane & 3) * 8);
                dst[0] = uint4{(uint32_t)t[0], (uint32_t)(t[0] >> 32), (uint32_t)t[1], (uint32_t)(t[1] >> 32)};
                dst[1] = uint4{(uint32_t)t[2], (uint32_t)(t[2] >> 32), (uint32_t)t[3], (uint32_t)(t[3] >> 32)};
            }
            // (LDS operations of one wave complete in order: the tiles are written before they are read below)
            const int nrows = nb >> 4, rest = nb & 15;
            Two rows[P];
            // rows of every group in lock step up to the longest set of the wave: no divergent control flow around
            // the LDS reads (a group that has run out of rows reads stale cells and its row is dropped by a select)
            int max_rows = 0;
#pragma unroll
            for (int j = 0; j < G; ++j) max_rows = max(max_rows, __builtin_amdgcn_readlane(nrows, j * KP));
            for (int r = 0; r < max_rows; ++r) {  // (a "#pragma unroll 2" stood here until round 6: the optimizer never applied it -- 28 warnings per build, the same code)
                const uint32_t *rowp = my_tile + r * STRIDE;
                uint32_t row[P];
#pragma unroll
                for (int c = 0; c < kRowTokens; c += 2) {
                    const uint32_t h0 = rowp[2 * c], h1 = rowp[2 * c + 2];
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const uint32_t k0 = sieve_key(h0, sp.a_lo[q], sp.b8[q]);
                        const uint32_t k1 = sieve_key(h1, sp.a_lo[q], sp.b8[q]);
                        row[q] = c == 0 ? min(k0, k1) : umin3(row[q], k0, k1);
                    }
                }
#pragma unroll
                for (int q = 0; q < P; ++q) rows[q].add(r < nrows ? tag16(row[q], (uint32_t)r) : kMaxHash);
            }
            if (__any(rest > 0)) {  // partial last rows (row index nrows <= 15: a full tile has no rest)
                uint32_t row[P];
#pragma unroll
                for (int q = 0; q < P; ++q) row[q] = kMaxHash;
                const uint32_t *rowp = my_tile + nrows * STRIDE;
                for (int c = 0; c < kRowTokens - 1; ++c) {
                    if (!__any(c < rest)) break;
                    const uint32_t h = rowp[2 * c];
#pragma unroll
                    for (int q = 0; q < P; ++q)
                        if (c < rest) row[q] = min(row[q], sieve_key(h, sp.a_lo[q], sp.b8[q]));
                }
#pragma unroll
                for (int q = 0; q < P; ++q)
                    if (rest > 0) rows[q].add(tag16(row[q], (uint32_t)nrows));
            }
            if (nb > 0) fail |= finish_block<P, STRIDE, 2, true>(rows, my_tile, pm, sp, res, (uint32_t)nrows, (uint32_t)rest);
        }
        const bool group_failed = (__ballot(fail && any_active) & group_lanes) != 0;
        if (has && k == 0) args.redo[set] = group_failed ? 1 : 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            if (has && sp.active[q] && !group_failed) {
                uint64_t v;
                if (args.init) {
                    const uint64_t iv = args.init[set * args.init_stride + k + q * KP];
                    v = end > beg ? (uint64_t)min((iv >> 32) ? kMaxHash : (uint32_t)iv, res[q]) : iv;  // empty set: state untouched
                } else {
                    v = end > beg ? res[q] : kMaxHash;
                }
                if (sizeof(OutT) == 4) v = v > kMaxHash ? kMaxHash : v;
                out[set * args.num_perm + k + q * KP] = (OutT)v;
            }
        }
    }
}

// ---- kernel B: few long sets, split over waves, combined with atomic min ---------------------
// out must already hold the initial state (init or 2^32-1).  grid.x = token slices of `slice`
// tokens over the flat token array; a slice may span several sets.
template <typename OutT>
__global__ void minhash_fill_state_kernel(const BulkArgs args) {
    const int64_t total = args.n_sets * (int64_t)args.num_perm;
    OutT *__restrict__ out = static_cast<OutT *>(args.out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t v = kMaxHash;
        if (args.init) {
            const int64_t set = i / args.num_perm;
            v = args.init[set * args.init_stride + (i - set * args.num_perm)];
        }
        if (sizeof(OutT) == 4) v = v > kMaxHash ? kMaxHash : v;
        out[i] = (OutT)v;
    }
}

template <int P, typename TokT, typename OutT>
__global__ __launch_bounds__(256) void minhash_split_kernel(const BulkArgs args, int64_t first_token,
                                                            int64_t total_tokens, int64_t slice) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_block = blockDim.x >> 6;
    __shared__ __attribute__((aligned(16))) uint32_t stage[4 * kStageWordsPerWave];  // one tile per wave
    uint32_t *lds = stage + wave * kStageWordsPerWave;
    Perms<P> pm, pm_biased;
    SievePerms<P> sp;
    int kidx[P];
    load_perms<P>(args, blockIdx.y * (kWave * P), lane, pm, pm_biased, sp, kidx);
    const TokT MHX_CONST_AS *hv = as_const(static_cast<const TokT *>(args.hv));
    const TokT *hv_vec = static_cast<const TokT *>(args.hv);
    const int64_t MHX_CONST_AS *offsets = as_const(args.offsets);
    OutT *__restrict__ out = static_cast<OutT *>(args.out);

    const int64_t n_slices = (total_tokens + slice - 1) / slice;
    const int64_t stride = (int64_t)gridDim.x * waves_per_block;
    SieveBackoff backoff;
    for (int64_t s = (int64_t)blockIdx.x * waves_per_block + wave; s < n_slices; s += stride) {
        const int64_t s_beg = first_token + s * slice;  // slices tile [first_token, first_token + total_tokens)
        const int64_t s_end = min(s_beg + slice, first_token + total_tokens);
        // first set whose range intersects [s_beg, s_end): largest i with start(i) <= s_beg
        int64_t set;
        if (args.offsets) {
            int64_t lo = 0, hi = args.n_sets;  // invariant: offsets[lo] <= s_beg < offsets[hi]
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (offsets[mid] <= s_beg) lo = mid; else hi = mid;
            }
            set = lo;
        } else {
            set = s_beg / args.fixed_len;
        }
        for (; set < args.n_sets; ++set) {
            const int64_t set_beg = args.offsets ? offsets[set] : set * args.fixed_len;
            const int64_t set_end = args.offsets ? offsets[set + 1] : set_beg + args.fixed_len;
            if (set_beg >= s_end) break;
            const int64_t beg = max(set_beg, s_beg), end = min(set_end, s_end);
            if (end <= beg) continue;
            uint32_t res[P];
            set_minima<P, TokT>(args, hv, hv_vec, beg, end, pm, pm_biased, sp, blockIdx.y * (kWave * P), lane, lds, backoff, res);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (kidx[p] < 0) continue;
                OutT *dst = out + set * args.num_perm + kidx[p];
                if (sizeof(OutT) == 8)
                    atomicMin(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)res[p]);
                else
                    atomicMin(reinterpret_cast<unsigned int *>(dst), res[p]);
            }
        }
    }
}

__global__ void minhash_merge_kernel(const uint64_t *__restrict__ x, const uint64_t *__restrict__ y,
                                     int64_t count, uint64_t *__restrict__ out) {
    // 16 B per lane per access, four accesses of each input in flight; HBM-bound elementwise min
    // (minhash.py:359)
    const int64_t n2 = count >> 1;
    const ulonglong2 *x2 = reinterpret_cast<const ulonglong2 *>(x);
    const ulonglong2 *y2 = reinterpret_cast<const ulonglong2 *>(y);
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(out);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 4 * stride) {
        ulonglong2 xa[4], ya[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride < n2) {  // non-temporal both ways, 32 workgroups per CU: tools/ubench_stream.hip
                xa[u].x = __builtin_nontemporal_load(&x2[i + u * stride].x);
                xa[u].y = __builtin_nontemporal_load(&x2[i + u * stride].y);
                ya[u].x = __builtin_nontemporal_load(&y2[i + u * stride].x);
                ya[u].y = __builtin_nontemporal_load(&y2[i + u * stride].y);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride >= n2) break;
            ulonglong2 r;
            r.x = xa[u].x < ya[u].x ? xa[u].x : ya[u].x;
            r.y = xa[u].y < ya[u].y ? xa[u].y : ya[u].y;
            __builtin_nontemporal_store(r.x, &o2[i + u * stride].x);
            __builtin_nontemporal_store(r.y, &o2[i + u * stride].y);
        }
    }
    if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        out[count - 1] = x[count - 1] < y[count - 1] ? x[count - 1] : y[count - 1];
}

// grid of 256-thread workgroups that is resident all at once (what the occupancy query says per CU, times the CUs),
// capped by the number of work items
unsigned resident_grid(mhx_ctx *ctx, const void *kernel, int64_t items) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 2;
    }
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(items, (int64_t)ctx->num_cus * per_cu));
}

// P permutations per lane in the sieve (and split) launch, PF in the full launch
template <int P, typename TokT, typename OutT, int PF = P>
int launch_typed(mhx_ctx *ctx, const BulkArgs &args, int64_t first_token, int64_t total_tokens, bool split) {
    const int kchunks = (args.num_perm + kWave * P - 1) / (kWave * P);
    // >> residency: the dispatcher evens out the tail.  Re-measured at steady clocks in round 4 (tools/experiments/
    // r04_steady_clock_revalidation.py): K = 128 dense 1.99 ms for 16 .. 128, 2.09 at 512; sets of 1..100 tokens 0.913 at 64,
    // 0.925 at 32, 0.929 at 128, 0.960 at 16; K = 256 (four permutations per lane: fewer waves resident, longer sets)
    // 3.89 at 64, 3.79 at 128, 3.81 at 256, 3.95 at 1024; K = 192 2.99 / 2.98 / 2.97 at 64 / 128 / 256
    const int blocks_per_cu = ctx->opt_blocks_per_cu > 0 ? (int)ctx->opt_blocks_per_cu : (P >= 3 ? 128 : 64);
    const int64_t max_blocks = (int64_t)ctx->num_cus * blocks_per_cu;
    if (!split) {
        const int64_t want = (args.n_sets + 3) / 4;
        dim3 grid((unsigned)std::max<int64_t>(1, std::min(want, max_blocks)), 1u);  // wave loops over kchunks
        if (args.path == 0) {
            // sieve launch (writes a flag per set), then the flagged sets (usually a handful: those launches read
            // n_sets bytes of flags and return)
            if (int rc = ctx->ensure_work()) return rc;
            MHX_HIP_CHECK(hipMemsetAsync(ctx->d_work, 0, 8 * sizeof(unsigned int), ctx->stream));
            BulkArgs sieve_args = args;
            sieve_args.sieve_hint = ctx->d_work;
            // (word 8 of d_work: not zeroed per call; minhash.adapt = 1 switches the device-side choice off; the packed kernels
            // of short signatures and counting runs -- whose counters are pinned by tests -- keep the one first launch)
            sieve_args.mode_word = ctx->opt_minhash_adapt == 1 || args.stats ? nullptr : ctx->d_work + 8;
            const BulkArgs &args_s = sieve_args;
            const bool plain = !args.init && !args.stats && args.alias_mask < 0;
            // short signatures: several sets per wave (kernel C) instead of a wave with most of its lanes idle
            // Several sets per wave (kernel C) instead of a wave with idle lanes or half-used registers: num_perm <= 32 one
            // permutation per lane on 8 / 16 / 32 lanes; 33 .. 64 three or four permutations on 16 lanes (four sets per
            // wave; K = 48: 1.45 -> 1.13 ms per 1M x 256, K = 64: 1.46 -> 1.40); 65 .. 96 three on 32 lanes (2.14 -> 1.96).
            // From 97 on one set per wave with its tokens on the scalar path is faster (K = 128: 2.19 against 2.42 ms, short
            // ragged sets 1.02 against 1.21): minhash.packed = 2 takes those through kernel C all the same (profiling).
            const int np = args.num_perm;
            const bool packable = !args.stats && args.alias_mask < 0 && ctx->opt_minhash_packed != 1;
            const bool packed = packable && P <= 2 && (np <= 96 || (ctx->opt_minhash_packed == 2 && np <= 128));
            if (packed) {
                const int kp = np <= 8 ? 8 : np <= 16 ? 16 : np <= 32 ? 32 : np <= 64 ? 16 : np <= 96 ? 32 : 64;
                const int64_t items = (args.n_sets + 64 / kp - 1) / (64 / kp);
                dim3 pgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>((items + 3) / 4, max_blocks)), 1u);
#define MHX_PACKED(KP_, P_) hipLaunchKernelGGL((minhash_packed_kernel<KP_, P_, TokT, OutT>), pgrid, dim3(256), 0, ctx->stream, args_s)
                if (np <= 8) MHX_PACKED(8, 1);
                else if (np <= 16) MHX_PACKED(16, 1);
                else if (np <= 32) MHX_PACKED(32, 1);
                else if (np <= 48) MHX_PACKED(16, 3);
                else if (np <= 64) MHX_PACKED(16, 4);
                else if (np <= 96) MHX_PACKED(32, 3);
                else MHX_PACKED(64, 2);
#undef MHX_PACKED
            } else {
            const bool fixed_rows = plain && !args.offsets && args.fixed_len % kRowTokens == 0;
            // 3 or 4 permutations per lane with at most 32 left for the last slot (K = 129..160, 193..224): the kernels whose lane
            // groups share that slot's rows (share_last_slot) instead of 64 - r lanes idling through every token
            const int last_slot = args.num_perm - (P - 1) * kWave;
            const bool share = P >= 3 && kchunks == 1 && args.share_last && last_slot > 0 && last_slot <= 32;
            const auto first_launches = [&](auto share_tag) {
                constexpr bool SH = decltype(share_tag)::value;
#define MHX_SIEVE(MODE_, SHAPE_) hipLaunchKernelGGL((minhash_bulk_kernel<P, TokT, OutT, MODE_, SHAPE_, SH>), grid, dim3(256), 0, ctx->stream, args_s)
                if (fixed_rows) MHX_SIEVE(MODE_SIEVE, SHAPE_PLAIN_FIXED_ROWS);  // whole 16-token rows: no tail code in the kernel
                else if (plain && !args.offsets) MHX_SIEVE(MODE_SIEVE, SHAPE_PLAIN_FIXED);
                else if (plain) MHX_SIEVE(MODE_SIEVE, SHAPE_PLAIN);
                else MHX_SIEVE(MODE_SIEVE, SHAPE_GENERAL);
                // the tie-tolerant first launch: works instead of the one above when the context's last call said so (see MODE_SIEVE_TIES)
                if (args_s.mode_word) {
                    if (fixed_rows) MHX_SIEVE(MODE_SIEVE_TIES, SHAPE_PLAIN_FIXED_ROWS);
                    else MHX_SIEVE(MODE_SIEVE_TIES, SHAPE_GENERAL);
                }
#undef MHX_SIEVE
            };
            if constexpr (P >= 3) {
                if (share) first_launches(std::true_type());
                else first_launches(std::false_type());
            } else {
                (void)share;
                first_launches(std::false_type());
            }
            }
            // the flagged sets: dedup sieve, then pair by pair what is still open (usually nothing: these launches read
            // n_sets bytes of flags and return).  A workgroup scans 64 flags at a time and strides over the flag
            // groups: a grid of one short-lived workgroup per group cost 60 us per step in workgroup launches alone
            const int64_t flag_groups = (args.n_sets + kWave - 1) / kWave;
            BulkArgs dedup = args;
            dedup.redo_match = 1;
            dedup.sieve_hint = sieve_args.sieve_hint;
            dedup.mode_word = packed ? nullptr : sieve_args.mode_word;
            if (args.n_sets <= 0xFFFFFFFFll) {
                dedup.pair_count = ctx->d_work + 4;
                dedup.pair_list = ctx->d_work + 16;
            }
            hipLaunchKernelGGL((minhash_bulk_kernel<PF, TokT, OutT, MODE_DEDUP, SHAPE_GENERAL>),
                               dim3(resident_grid(ctx, (const void *)minhash_bulk_kernel<PF, TokT, OutT, MODE_DEDUP, SHAPE_GENERAL>, flag_groups)),
                               dim3(256), 0, ctx->stream, dedup);
            BulkArgs rest = dedup;
            rest.redo_match = 2;
            hipLaunchKernelGGL((minhash_bulk_kernel<PF, TokT, OutT, MODE_FULL, SHAPE_GENERAL>),
                               dim3(resident_grid(ctx, (const void *)minhash_bulk_kernel<PF, TokT, OutT, MODE_FULL, SHAPE_GENERAL>, flag_groups)),
                               dim3(256), 0, ctx->stream, rest);
        } else {
            BulkArgs all = args;
            all.redo = nullptr;  // every set
            hipLaunchKernelGGL((minhash_bulk_kernel<PF, TokT, OutT, MODE_FULL, SHAPE_GENERAL>), grid, dim3(256), 0, ctx->stream, all);
        }
    } else {
        const int64_t total_out = args.n_sets * (int64_t)args.num_perm;
        const int64_t fill_blocks = std::max<int64_t>(1, std::min<int64_t>((total_out + 255) / 256, max_blocks));
        hipLaunchKernelGGL((minhash_fill_state_kernel<OutT>), dim3((unsigned)fill_blocks), dim3(256), 0,
                           ctx->stream, args);
        // slice length: whole 256-token sieve blocks, about 8 slices (waves) per SIMD; short inputs
        // keep 64-token slices so that a single 50k-token update_batch still spreads over the chip
        const int64_t waves = (int64_t)ctx->num_cus * 32;
        int64_t slice = (total_tokens + waves - 1) / waves;
        slice = slice > 128 ? (slice + 255) / 256 * 256 : std::max<int64_t>(64, (slice + 15) / 16 * 16);
        const int64_t n_slices = (total_tokens + slice - 1) / slice;
        const int64_t want = (n_slices + 3) / 4;
        dim3 grid((unsigned)std::max<int64_t>(1, std::min(want, max_blocks)), (unsigned)kchunks);
        hipLaunchKernelGGL((minhash_split_kernel<P, TokT, OutT>), grid, dim3(256), 0, ctx->stream, args,
                           first_token, total_tokens, slice);
    }
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

template <typename TokT, typename OutT>
int launch_p(mhx_ctx *ctx, const BulkArgs &args, int64_t first_token, int64_t total_tokens, bool split) {
    // P permutations per lane: 1 for K <= 64
    if (args.num_perm <= 64) return launch_typed<1, TokT, OutT>(ctx, args, first_token, total_tokens, split);
    // Beyond 64: 2, 3 or 4 permutations per lane in the sieve launch -- whichever walks the fewest 64-permutation slots over its
    // ceil(K / 64P) passes (a pass costs its P slots whether they are full or not; the passes re-read the set from L2).  Ties go to the
    // larger P (fewer passes: K = 256 4.00 -> 3.90 ms with four per lane).  K = 129..192: three (one pass, every lane busy up to 192;
    // four per lane left a quarter of the lanes idle there); 193..256: four; 257..384: three, twice (round 4; until then two per lane in three
    // passes: the same 6 slots, one more pass over the set); 385..512: four, twice; 513..576: three, three times.  The full launch stays at two (with four
    // it would need 220 VGPRs).  Four per lane for uint64 tokens only (uint32: a row per chunk, two rows per loop body: 207 VGPRs;
    // three: 121).  Option minhash.p3 = 1 takes the three-per-lane kernels out (A/B).
    const auto slots = [&](int p) { return (args.num_perm + kWave * p - 1) / (kWave * p) * p; };
    const bool may3 = !split && ctx->opt_minhash_p3 != 1, may4 = !split && sizeof(TokT) == 8;
    int best = 2;
    if (may3 && slots(3) <= slots(best)) best = 3;
    if (may4 && slots(4) <= slots(best)) best = 4;
    if (best == 3) return launch_typed<3, TokT, OutT, 2>(ctx, args, first_token, total_tokens, split);
    if (best == 4) return launch_typed<4, TokT, OutT, 2>(ctx, args, first_token, total_tokens, split);
    return launch_typed<2, TokT, OutT>(ctx, args, first_token, total_tokens, split);
}

}  // namespace

int launch_minhash_bulk(mhx_perm *perm, const void *d_hv, int hv_dtype, const int64_t *d_offsets,
                        int64_t fixed_len, int64_t n_sets, int64_t total_tokens,
                        const uint64_t *d_init, int64_t init_stride, void *d_out, int out_dtype, int64_t first_token) {
    // the sets' tokens lie in d_hv[first_token, total_tokens): first_token is 0 for a whole corpus (tokens
    // in front of offsets[0] are then merely never touched) and the window start for a piece of one
    mhx_ctx *ctx = perm->ctx;
    if (n_sets == 0) return MHX_OK;
    total_tokens -= first_token;  // from here on: the number of tokens in the window
    BulkArgs args;
    args.hv = d_hv;
    args.offsets = d_offsets;
    args.fixed_len = fixed_len;
    args.n_sets = n_sets;
    args.a = perm->d_a;
    args.b = perm->d_b;
    args.num_perm = perm->num_perm;
    args.path = (int32_t)ctx->opt_minhash_path;
    args.stats = ctx->d_stats;
    args.redo = nullptr;
    args.redo_match = 1;
    args.sieve_hint = nullptr;
    args.mode_word = nullptr;
    args.alias_mask = ctx->opt_minhash_alias;
    args.pair_count = nullptr;
    args.pair_list = nullptr;
    // the one-set-ahead warm-up load: 1 (default) = where it pays -- CSR sets and fixed-length sets shorter than 256 tokens (3-6 % at
    // steady clocks: 64 / 100 / 128-token sets 1.72 -> 1.63 / 1.21 -> 1.15 / 1.27 -> 1.24 ms, ragged 32..480 1.276 -> 1.230); on dense
    // sets of 256 tokens and more it buys nothing (2.09 against 2.13 ms on the box of that sweep, 512 tokens 2.016 against 2.017) and
    // reads a third of the corpus twice (lines evicted from the 4 MB L2 before use: 1.23x the algorithmic bytes, round 2-4 profiles).
    // 0 = never, 2 = always.
    args.prefetch = ctx->opt_minhash_prefetch == 2 || (ctx->opt_minhash_prefetch == 1 && (d_offsets != nullptr || fixed_len < 256));
    args.ties = ctx->opt_minhash_ties != 1;
    args.share_last = ctx->opt_minhash_share != 1;
    args.init = d_init;
    args.init_stride = init_stride;
    args.out = d_out;
    // Few sets with long token lists: split sets over waves (atomic combine); else wave per set.
    const int64_t waves_avail = (int64_t)ctx->num_cus * 16;
    bool split = n_sets < waves_avail && total_tokens > n_sets * 128 && total_tokens >= 1024;
    if (ctx->opt_minhash_split == 1) split = false;
    if (ctx->opt_minhash_split == 2) split = total_tokens > 0;
    ctx->redo_sets = 0;
    if (!split && args.path == 0) {
        if (int rc = ctx->ensure_redo(n_sets)) return rc;
        args.redo = ctx->d_redo;
        ctx->redo_sets = n_sets;  // (mhx_ctx_minhash_flags)
    }
    if (hv_dtype == MHX_U64 && out_dtype == MHX_U64) return launch_p<uint64_t, uint64_t>(ctx, args, first_token, total_tokens, split);
    if (hv_dtype == MHX_U64 && out_dtype == MHX_U32) return launch_p<uint64_t, uint32_t>(ctx, args, first_token, total_tokens, split);
    if (hv_dtype == MHX_U32 && out_dtype == MHX_U64) return launch_p<uint32_t, uint64_t>(ctx, args, first_token, total_tokens, split);
    if (hv_dtype == MHX_U32 && out_dtype == MHX_U32) return launch_p<uint32_t, uint32_t>(ctx, args, first_token, total_tokens, split);
    return fail(MHX_ERR_INVALID, "unknown hv_dtype/out_dtype (%d, %d)", hv_dtype, out_dtype);
}

int launch_minhash_merge(mhx_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, int64_t count,
                         uint64_t *d_out) {
    if (count == 0) return MHX_OK;
    const int64_t want = ((count >> 1) + 1023) / 1024;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)ctx->num_cus * 32));
    hipLaunchKernelGGL(minhash_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_x, d_y,
                       count, d_out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx

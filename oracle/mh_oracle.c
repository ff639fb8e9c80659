/*
 * oracle/mh_oracle.c -- CPU restatement of the datasketch bulk-MinHash hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under datasketch_amd/ may link, import or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 * against golden vectors produced by importing the real reference from
 * /root/reference (oracle/gen_golden.py, fixtures in tests/golden/) and against
 * the known-answer vector of the reference's own test (test/test_minhash.py:109-115).
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference/).  Plain scalar C, no SIMD, no threads: one obvious loop per
 * reference expression so it can be audited against the numpy source by eye.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off, see oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* datasketch/minhash.py:30-31  _mersenne_prime = 2^61-1, _max_hash = 2^32-1 */
static const uint64_t MERSENNE_P = (((uint64_t)1) << 61) - 1;
static const uint64_t MAX_HASH = (((uint64_t)1) << 32) - 1;

/*
 * One permuted hash value.
 * datasketch/minhash.py:295-296 (update_batch CPU body) and :223 (update):
 *     phv = (hv * a + b) % _mersenne_prime ; phv = bitwise_and(phv, _max_hash)
 * numpy evaluates hv*a+b in uint64, i.e. modulo 2^64 -- C unsigned arithmetic
 * has the same wrap, so the expression is transcribed literally.
 */
static inline uint64_t permute(uint64_t hv, uint64_t a, uint64_t b) {
    uint64_t s = hv * a + b; /* wraps mod 2^64 exactly like numpy uint64 */
    return (s % MERSENNE_P) & MAX_HASH;
}

/*
 * Bulk MinHash over a ragged (CSR) corpus of pre-hashed tokens.
 *
 * Restates MinHash.generator / MinHash.bulk (datasketch/minhash.py:464-522):
 * every set starts from a copy of the prototype state (init, or all
 * _max_hash when init == NULL; minhash.py:167-168) and receives one
 * update_batch (minhash.py:293-297).  An empty set leaves the state untouched
 * (minhash.py:265-266).
 *
 *   hv       [offsets[n_sets]]  token hash values (output of hashfunc, uint64)
 *   offsets  [n_sets+1]         CSR row pointers
 *   a, b     [num_perm]         permutations[0], permutations[1]
 *   init     NULL | [num_perm] (init_stride=0) | [n_sets,num_perm] (init_stride=num_perm)
 *   out      [n_sets, num_perm] uint64
 */
ORACLE_API void oracle_minhash_bulk(const uint64_t *hv, const int64_t *offsets, int64_t n_sets,
                                    const uint64_t *a, const uint64_t *b, int32_t num_perm,
                                    const uint64_t *init, int64_t init_stride, uint64_t *out) {
    for (int64_t i = 0; i < n_sets; ++i) {
        uint64_t *row = out + i * (int64_t)num_perm;
        for (int32_t k = 0; k < num_perm; ++k)
            row[k] = init ? init[i * init_stride + k] : MAX_HASH;
        for (int64_t t = offsets[i]; t < offsets[i + 1]; ++t) {
            const uint64_t h = hv[t];
            for (int32_t k = 0; k < num_perm; ++k) {
                const uint64_t p = permute(h, a[k], b[k]);
                if (p < row[k]) row[k] = p; /* np.minimum, minhash.py:297 */
            }
        }
    }
}

/* MinHash.merge (minhash.py:337-359) / union (:411-462) on whole matrices:
 * elementwise minimum of two [n, num_perm] signature matrices. */
ORACLE_API void oracle_minhash_merge(const uint64_t *x, const uint64_t *y, int64_t count,
                                     uint64_t *out) {
    for (int64_t i = 0; i < count; ++i) out[i] = x[i] < y[i] ? x[i] : y[i];
}

/*
 * WeightedMinHashGenerator.minhash_many (datasketch/weighted_minhash.py:161-247)
 * for a CSR matrix whose stored values are all non-zero.
 *
 * The natural log of the data (weighted_minhash.py:212) is an INPUT here
 * (log_data[nnz], float32): numpy's float32 log is not correctly rounded and
 * is CPU-dispatch dependent, so the caller computes it with numpy -- the same
 * binary the reference uses -- and everything after it is plain IEEE float32:
 *
 *   t    = floor(log_data / rs + betas)               :216
 *   ln_y = (t - betas + 1) * rs                       :217
 *   ln_a = ln_cs - ln_y                               :218
 *   per row, per sample: first-index argmin of ln_a   :229  (np.argmin)
 *   hashvalues[:,0] = column index at argmin, [:,1] = t there   :233-239
 *
 * Compiled with -ffp-contract=off so that no a*b+c is fused (numpy never fuses).
 *
 *   rs, ln_cs, betas  [sample_size, dim] float32, row-major (as the reference holds them)
 *   out               [n_rows, sample_size, 2] int64
 *   nonempty          [n_rows] uint8: 0 where the row has no stored value (reference
 *                     returns None for those rows, :242-247)
 */
ORACLE_API void oracle_weighted_minhash_many(const int64_t *indptr, const int32_t *indices,
                                             const float *log_data, int64_t n_rows,
                                             const float *rs, const float *ln_cs,
                                             const float *betas, int32_t sample_size,
                                             int32_t dim, int64_t *out, uint8_t *nonempty) {
    for (int64_t d = 0; d < n_rows; ++d) {
        const int64_t beg = indptr[d], end = indptr[d + 1];
        nonempty[d] = (uint8_t)(end > beg);
        int64_t *row = out + d * (int64_t)sample_size * 2;
        if (end == beg) {
            memset(row, 0, sizeof(int64_t) * 2 * (size_t)sample_size);
            continue;
        }
        for (int32_t i = 0; i < sample_size; ++i) {
            const float *r_i = rs + (int64_t)i * dim;
            const float *c_i = ln_cs + (int64_t)i * dim;
            const float *b_i = betas + (int64_t)i * dim;
            float best = 0.0f, best_t = 0.0f;
            int32_t best_k = -1;
            for (int64_t j = beg; j < end; ++j) {
                const int32_t col = indices[j];
                const float q = log_data[j] / r_i[col];      /* :216 */
                const float tt = floorf(q + b_i[col]);       /* :216 */
                const float u = tt - b_i[col];               /* :217 */
                const float v = u + 1.0f;                    /* :217 */
                const float ln_y = v * r_i[col];             /* :217 */
                const float ln_a = c_i[col] - ln_y;             /* :218 */
                /* np.argmin: first minimum wins; NaN propagates as the minimum */
                if (best_k < 0 || ln_a < best || (isnan(ln_a) && !isnan(best))) {
                    best = ln_a;
                    best_t = tt;
                    best_k = col;
                }
            }
            row[2 * i + 0] = best_k;
            row[2 * i + 1] = (int64_t)best_t; /* float -> int64 assignment, :236-239 */
        }
    }
}

/*
 * bBitMinHash packing (datasketch/b_bit_minhash.py:37-38, 78-101, 147-172).
 *   hashvalues & ((1<<b)-1)  -> uint32, then n = 64/slot values per uint64 block,
 *   value j of a block at bit (n-1-j)*slot.
 * in   [n_sigs, num_perm] uint64 signatures
 * out  [n_sigs, num_blocks] uint64, num_blocks = ceil(num_perm / n)
 */
static int slot_size_for(int b) { /* b_bit_minhash.py:147-160 */
    if (b == 1) return 1;
    if (b == 2) return 2;
    if (b <= 4) return 4;
    if (b <= 8) return 8;
    if (b <= 16) return 16;
    return 32;
}

ORACLE_API int32_t oracle_bbit_num_blocks(int32_t num_perm, int32_t b) {
    const int n = 64 / slot_size_for(b);
    return (num_perm + n - 1) / n;
}

ORACLE_API void oracle_bbit_pack(const uint64_t *sig, int64_t n_sigs, int32_t num_perm, int32_t b,
                                 uint64_t *out) {
    const int slot = slot_size_for(b);
    const int n = 64 / slot;
    const int32_t nb = (num_perm + n - 1) / n;
    const uint64_t bmask = b >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << b) - 1);
    for (int64_t i = 0; i < n_sigs; ++i) {
        for (int32_t blk = 0; blk < nb; ++blk) {
            uint64_t word = 0;
            for (int j = 0; j < n; ++j) {
                const int32_t k = blk * n + j;
                if (k >= num_perm) break;
                const uint64_t hvb = (uint32_t)(sig[i * (int64_t)num_perm + k] & bmask); /* :38 */
                word |= hvb << ((n - 1 - j) * slot);                                     /* :97 */
            }
            out[i * (int64_t)nb + blk] = word;
        }
    }
}

/*
 * MinHashLSH band keys (datasketch/lsh.py:199, 344, 537-538): the key of band i is
 * bytes(hashvalues[i*r:(i+1)*r].byteswap().data), i.e. each uint64 stored big-endian.
 * in [n_sigs, num_perm] uint64 -> out [n_sigs, bands*r] uint64 whose in-memory bytes are the keys.
 */
ORACLE_API void oracle_band_keys(const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                                 int32_t bands, int32_t r, uint64_t *out) {
    const int32_t w = bands * r;
    for (int64_t i = 0; i < n_sigs; ++i)
        for (int32_t k = 0; k < w; ++k)
            out[i * (int64_t)w + k] = __builtin_bswap64(sig[i * (int64_t)num_perm + k]);
}

/*
 * LeanMinHash.serialize payload for a whole matrix (datasketch/lean_minhash.py:126-175):
 * per signature  <byteorder> q (seed) i (num_perm) {num_perm}I  = 12 + 4*num_perm bytes,
 * little-endian variant ('<'; no padding between q and i in struct's standard mode).
 */
ORACLE_API void oracle_lean_serialize_le(const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                                         int64_t seed, uint8_t *out) {
    const int64_t rec = 12 + 4 * (int64_t)num_perm;
    for (int64_t i = 0; i < n_sigs; ++i) {
        uint8_t *p = out + i * rec;
        memcpy(p, &seed, 8);
        memcpy(p + 8, &num_perm, 4);
        for (int32_t k = 0; k < num_perm; ++k) {
            const uint32_t v = (uint32_t)sig[i * (int64_t)num_perm + k];
            memcpy(p + 12 + 4 * (int64_t)k, &v, 4);
        }
    }
}

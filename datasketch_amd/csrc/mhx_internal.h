// mhx_internal.h -- shared plumbing of libmhx (context, error reporting, launch helpers).
// Product code: nothing here may reference oracle/.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <unordered_map>

#include "mhx.h"

namespace mhx {

// thread-local last-error string (mhx_last_error)
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
void forgive();  // a failure the caller tolerates (an optional buffer that could not be had): the message of a call that succeeds is empty

#define MHX_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            return ::mhx::fail(_e == hipErrorOutOfMemory ? MHX_ERR_OOM : MHX_ERR_HIP,            \
                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                               __LINE__);                                                        \
        }                                                                                        \
    } while (0)

// internal "signature type" of launch_lsh_sort_bands: the matrix holds band digests already ([n, bands] uint64)
constexpr int kSigDigests = 2;
constexpr int kSigDigestsBM = 3;  // ... band-major ([bands, n])

// mhx_ctx::d_work: 16 counter words, then the list of sets the second MinHash launch leaves to the pairwise one
constexpr unsigned int kPairListCap = 16384;
constexpr size_t kWorkBytes = 64 + sizeof(unsigned int) * kPairListCap;

// serialise the calls on one context (see mhx_ctx::mu)
#define MHX_GUARD(ctxp) std::lock_guard<std::recursive_mutex> _mhx_guard((ctxp)->mu)

#define MHX_REQUIRE(cond, ...)                                       \
    do {                                                             \
        if (!(cond)) return ::mhx::fail(MHX_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// every device allocation of the library (mhx_api.hip): hipMalloc / hipFree, or guard-paged mappings in guard mode
hipError_t dev_malloc(void **p, size_t bytes, bool caller = false);
hipError_t dev_free(void *p);
bool guard_mode();

}  // namespace mhx

// Opaque handle layouts (C linkage names are declared in mhx.h).
struct mhx_ctx {
    // One stream, one set of staging buffers: calls on a context are serialised (ctypes releases the GIL, so
    // two Python threads can be inside libmhx at once).  Recursive: host entry points call each other.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int64_t lds_per_block = 64 << 10;  // what a workgroup may ask for (160 KB on gfx950)
    int64_t hbm_bytes = 0;
    char name[128] = {0};
    // grow-only scratch used by the host entry points (device staging of inputs/outputs)
    void *scratch[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[5] = {0, 0, 0, 0, 0};
    // options (mhx_ctx_set_option)
    int64_t opt_minhash_path = 0;   // 0 auto (sieve + fallbacks), 1 exact fold everywhere, 2 fast fold (+exact redo)
    int64_t opt_minhash_packed = 0; // 0 auto (several sets per wave when num_perm <= 96), 1 always one set per wave, 2 several sets per wave up to 128
    int64_t opt_minhash_adapt = 0;  // 0 auto (the first launch is the tie-tolerant one when the previous call's sets mostly defeated the one-candidate proof), 1 = never
    int64_t opt_minhash_share = 0;  // 0 auto (lane groups share the rows of a last slot with <= 32 permutations: K = 129..160, 193..224), 1 = off
    int64_t opt_minhash_p3 = 0;     // 0 auto (three per lane in the sieve launch where that walks the fewest slots: 129 .. 192, 257 .. 384, 513 .. 576), 1 = never three
    int64_t opt_minhash_ties = 0;   // 0 auto (the second launch tries the tie-tolerant sieve before the dedup pass), 1 dedup pass only
    int64_t opt_minhash_split = 0;  // 0 auto, 1 force wave-per-set, 2 force split-sets (atomic combine)
    int64_t opt_blocks_per_cu = 0;  // 0 auto
    int64_t opt_minhash_prefetch = 1; // warm L2 with the next set's tokens (vector load per set): 1 auto (CSR, or fixed length < 256), 0 never, 2 always
    int64_t opt_minhash_alias = -1; // profiling only: >= 0 makes set i read the tokens of set (i & mask)
    int64_t opt_weighted_path = 0;  // 0 auto (dense rows: bound-ordered walk; CSR: reciprocal-multiply quotient + row blocks), 1 IEEE division for every element, 2 every element evaluated (dense rows compacted to CSR: the round-2 path)
    int64_t opt_weighted_min_dim = 0; // dense walk: rows of at least this many columns (a multiple of 4, <= 4096) go to the wave / fetcher-walker kernels; 0 auto
    int64_t opt_weighted_rescue = 0; // dense walk, one wave per row: a walk's last lanes get the whole wave each (walk_rescue) when at most this many are left; 0 auto (8), < 0 never
    int64_t opt_weighted_plan = 0;   // walk plan + tables: 0 = one launch (every workgroup plans, then sorts its sample's list if need be), 1 = round 3's two launches
    int64_t opt_weighted_kernel = 0; // dense walk: 0 auto (one wave per row where the shape allows, two chunks of samples walked as one stream), 1 = one workgroup per row always, 2 = one wave per row, chunk after chunk
    int64_t opt_weighted_debug = 0;  // profiling only (results are wrong): 1 = rows staged and scanned, not walked; 2 = staged without the scan
    int64_t opt_weighted_split = 0;  // dense walk kernel, rows evaluated entry by entry: 0 = the waves of a workgroup that share a chunk of samples split the list, 1 = one wave per chunk
    int64_t opt_weighted_tail = 0;   // walk plan: 0 = the cut with the smallest estimated cost, 1 .. 5 = that entry of kCutTail (profiling)
    int64_t opt_weighted_direct = 0; // walk kernel: rows storing at most this many per mille of the columns are evaluated entry by entry; 0 auto
    int64_t opt_lsh_sort = 0;       // mhx_lsh_sort_bands: 0 auto (two-pass bucketing, radix sort when a bin would overflow), 1 radix sort
    int64_t opt_lsh_gather = 0;     // mhx_lsh_sort_bands: 1 = gather the full digests after the sort (the fallback path) even when they could ride along
    int64_t opt_lsh_sort_bits = 0;  // mhx_lsh_sort_bands: bits of (band, digest) the radix sort orders by; 0 = from n
    int64_t opt_lsh_prehash = 0;    // mhx_lsh_sort_bands on a signature matrix: 0 auto (band digests first, then the bucketing), 1 = hash inside the scatter pass (A/B)
    int64_t opt_lsh_chunk = 0;      // bucketing: rows per thread of a scatter pass over a unit-stride source: 0 auto (16), 8 = eight (A/B)
    int64_t opt_lsh_bigbins = 0;    // bucketing: 0 auto (2.56M .. 10.2M rows: one scatter level into 1024 big bins + the big bin pass), 1 never, 2 whenever there are at least 4 bins (tests)
    int64_t opt_lsh_team = 0;       // bucketing, unit-stride sources: 0 auto (one team of 1024 threads x 8 rows per workgroup), 256 = teams of 256 x 16 (until round 6; A/B, tests)
    int64_t opt_lsh_levels = 0;     // bucketing: 0 auto (two scatter levels beyond 2^10 bins per band), 2 = two levels whenever there are at least 4 bins
    int64_t opt_pack_fused = 0;     // mhx_bbit_pack_band_digests_dev: 0 auto (one read of the matrix where the shape allows), 1 = always the two kernels
    int64_t opt_weighted_refill = 0; // one-wave-per-row walk, 4096-column rows: 0 auto (non-temporal row loads; values in: also the refill right after staging, chunk after chunk), 1 = round 4 (plain loads, refill behind the walk), 2 / 3 = force non-temporal / + early refill
    int64_t opt_host_chunk_bytes = 0;  // mhx_minhash_bulk: bytes per pipelined piece; 0 auto (96 MiB, inputs > 256 MiB), < 0 never pipeline

    // copy streams of the pipelined host entry point (created on first use)
    hipStream_t copy_in = nullptr;
    hipStream_t copy_out = nullptr;
    int ensure_copy_streams();

    // device counters of the MinHash kernels (mhx_ctx_counters); nullptr until counting is enabled
    unsigned long long *d_stats = nullptr;
    // redo flags of the wave-per-set MinHash launches: one byte per set (grow-only)
    uint8_t *d_redo = nullptr;
    int64_t redo_capacity = 0;
    int64_t redo_sets = 0;          // sets whose flags the last MinHash launch sequence wrote (0: that call kept no flags)
    int ensure_redo(int64_t n_sets);
    // the sieve launch's running failure counts ([0], [1]), the number of sets left to the pairwise launch ([4]) -- zeroed
    // per call -- and, from word 16 on, the list of those sets (kPairListCap entries)
    unsigned int *d_work = nullptr;
    int ensure_work();

    int ensure_scratch(int slot, size_t bytes);
    int activate() const;
};

struct mhx_perm {
    mhx_ctx *ctx = nullptr;
    int32_t num_perm = 0;
    uint64_t *d_a = nullptr;  // [K]
    uint64_t *d_b = nullptr;  // [K]
};

struct mhx_wgen {
    mhx_ctx *ctx = nullptr;
    int32_t sample_size = 0;
    int32_t dim = 0;
    // parameters transposed to [dim][5][S_pad] words (double 1/r, then r, ln_c, beta) so that one
    // column's samples are contiguous across lanes
    float *d_params = nullptr;
    int32_t s_pad = 0;
    // every r is finite with 2^-40 <= |r| <= 2^40: the reciprocal-multiply quotient is proven exact
    // (weighted_kernels.hip); otherwise every element takes the IEEE division
    bool table_fast = false;
    // {r, ln_c, beta, 0} [dim][S_pad]: one 16-byte load per lane for a wave-uniform column
    float *d_aos = nullptr;
    // the walk kernel's tables (weighted_kernels.hip): per 64-sample chunk, the columns in the order of a lower bound of
    // ln_a -- {bound, r, ln_c, beta} and the column, [S_pad / 64][dim][64] each -- and the device-resident plan they
    // were built for.  walk_ok: r > 0, ln_c and beta finite everywhere (the bound's monotonicity) and dim <= 16384 (LDS)
    float *d_walk_a = nullptr;
    uint32_t *d_walk_c = nullptr;
    void *d_walk_plan = nullptr;  // two WalkPlan records: the one the last call wrote (plan_index) and the one the next call will write
    int plan_index = 0;
    bool walk_ok = false;
};

struct mhx_event {
    mhx_ctx *ctx = nullptr;
    hipEvent_t ev = nullptr;
};

namespace mhx {

// kernels' host-side launchers (defined in the .hip files)
int launch_minhash_bulk(mhx_perm *perm, const void *d_hv, int hv_dtype, const int64_t *d_offsets,
                        int64_t fixed_len, int64_t n_sets, int64_t total_tokens,
                        const uint64_t *d_init, int64_t init_stride, void *d_out, int out_dtype,
                        int64_t first_token = 0);
int launch_sha1_tokens(mhx_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_offsets, int64_t n_tokens,
                       int out_dtype, void *d_out);
int launch_minhash_merge(mhx_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, int64_t count,
                         uint64_t *d_out);
int launch_weighted(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices,
                    const float *d_values, int values_are_logs, int64_t n_rows, int64_t nnz,
                    int64_t *d_out, uint8_t *d_nonempty);
int launch_weighted_log(mhx_ctx *ctx, const float *d_x, int64_t n, float *d_out);
int launch_wgen_transpose(mhx_wgen *gen, const float *d_rs, const float *d_lncs, const float *d_betas);
int launch_bbit_pack(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b,
                     uint64_t *d_out);
int launch_band_keys(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands,
                     int32_t r, uint64_t *d_out);
int launch_band_digests(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands, int32_t r,
                        uint64_t *d_out, int layout = 0);
int launch_bbit_digest_fused(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b, int32_t bands,
                             int32_t r, uint64_t *d_blocks, uint64_t *d_digests, int layout, bool *done);
int launch_jaccard_pairs(mhx_ctx *ctx, const void *d_a, const void *d_b, int sig_dtype, int32_t k, const int64_t *d_pairs,
                         int64_t m, int32_t *d_counts);
int launch_weighted_dense(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows, int64_t *d_out,
                          uint8_t *d_nonempty);
int launch_lsh_candidate_pairs(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n,
                               int32_t bands, int64_t *d_pairs, int64_t capacity, int64_t *n_pairs, int64_t *n_raw);
int launch_lsh_sort_bands(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands, int32_t r,
                          uint64_t *d_sorted_digests, uint32_t *d_sorted_rows);
int launch_lsh_query(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n, int32_t bands,
                     int32_t r, const void *d_q_sig, const void *d_idx_sig, int sig_dtype, int32_t k, int64_t m,
                     int64_t *d_pairs, int64_t capacity, int64_t *n_pairs);
int launch_bbit_jaccard(mhx_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, int32_t k, int32_t b, const int64_t *d_pairs,
                        int64_t m, int32_t *d_counts);
int launch_lean_serialize(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int64_t seed, int big_endian,
                          uint8_t *d_out);
int launch_lean_deserialize(mhx_ctx *ctx, const uint8_t *d_records, int64_t n, int32_t k, int big_endian, int sig_dtype, void *d_sig,
                            int64_t *d_seeds, unsigned int *d_bad);
int launch_bbit_unpack(mhx_ctx *ctx, const uint64_t *d_blocks, int64_t n, int32_t k, int32_t b, uint32_t *d_out);

int bbit_slot_size(int b);

}  // namespace mhx

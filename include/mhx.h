/*
 * mhx.h -- C ABI of libmhx, the MI355X-native MinHash signature engine.
 *
 * This is the drop-in boundary for ekzhu/datasketch's bulk-hashing hot path.  It is what a
 * ctypes / cffi stub inside the reference would bind in place of its optional CuPy branch
 * (reference datasketch/minhash.py:281-291; binding shown in INTEGRATION.md).  Plain pointers
 * and sizes only: no C++ types, no torch types, no exceptions across the boundary.
 *
 * Conventions
 *   - every function returns an int status (MHX_OK == 0); mhx_last_error() gives the
 *     thread-local message of the last failing call.
 *   - "host" entry points take caller-owned host buffers, block until the result is in host
 *     memory, and never retain the pointers.  "_dev" entry points take device pointers
 *     (from mhx_dev_alloc or any other HIP allocation on the same device), enqueue on the
 *     context's stream and return without synchronising.
 *   - a context owns one device + one HIP stream; calls on one context (and on the perm / wgen /
 *     comm handles created from it) are serialised by a mutex inside the context, so concurrent
 *     callers are safe but do not overlap; distinct contexts are independent (one context per
 *     GPU / per process is the multi-GPU model).
 *   - device buffers handed to "_dev" entry points need NO slack: a kernel reads and writes only the bytes
 *     the argument list describes ([n_sigs, num_perm] elements, offsets[n_sets] tokens ...), and pointers need
 *     only the natural alignment of their element type (wider loads are used where the pointer allows them).
 *     tests/test_guard_pages.py holds the kernels to this with mhx_debug_guard_alloc (an unmapped page right
 *     behind -- or in front of -- every buffer).
 *   - citations "ref:" are paths in the reference repository (ekzhu/datasketch v1.10.0).
 */
#ifndef MHX_H_
#define MHX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHX_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------------------------- */
enum {
    MHX_OK = 0,
    MHX_ERR_NO_DEVICE = 1,   /* no usable HIP device: maps to the reference's RuntimeError for
                                gpu_mode='always' (ref: datasketch/minhash.py:272-275)          */
    MHX_ERR_INVALID = 2,     /* bad argument / shape: maps to ValueError                          */
    MHX_ERR_HIP = 3,         /* a HIP runtime call failed                                         */
    MHX_ERR_OOM = 4,         /* device allocation failed                                          */
    MHX_ERR_UNSUPPORTED = 5, /* valid request outside what this build implements                  */
    MHX_ERR_COMM = 6         /* RCCL failure                                                      */
};

/* element types of token / signature buffers */
enum { MHX_U64 = 0, MHX_U32 = 1 };

typedef struct mhx_ctx mhx_ctx;     /* device + stream + scratch                                  */
typedef struct mhx_perm mhx_perm;   /* MinHash permutations (a[K], b[K]) resident on the device   */
typedef struct mhx_wgen mhx_wgen;   /* WeightedMinHashGenerator parameters resident on the device */
typedef struct mhx_event mhx_event; /* HIP event on the context's stream                          */
typedef struct mhx_comm mhx_comm;   /* RCCL communicator (one rank per context)                   */

/* ---- library / device --------------------------------------------------------------------- */
MHX_API const char *mhx_last_error(void);
MHX_API const char *mhx_version(void);

/* Number of usable devices.  Replaces ref: datasketch/minhash.py:38-48 (_gpu_available). */
MHX_API int mhx_device_count(int *count);

MHX_API int mhx_ctx_create(int device, mhx_ctx **ctx);
MHX_API int mhx_ctx_destroy(mhx_ctx *ctx);
MHX_API int mhx_ctx_synchronize(mhx_ctx *ctx);
/* Free the grow-only device staging buffers of the host entry points (they are re-created on demand). */
MHX_API int mhx_ctx_release_scratch(mhx_ctx *ctx);
/* name: caller buffer (may be NULL); cus: compute units; hbm_bytes: total device memory */
MHX_API int mhx_ctx_device_info(mhx_ctx *ctx, char *name, int name_len, int *cus, int64_t *hbm_bytes);
/* Tuning / test knobs: ("minhash.path", 0 = auto: sieve with dedup / pairwise fallback launches,
 * 1 = exact fold for every pair, 2 = fast fold with exact redo), ("minhash.split", 0 auto,
 * 1 wave per set, 2 split sets over waves), ("minhash.packed", 0 auto: several sets per wave when
 * num_perm <= 96, 1 = always one set per wave, 2 = several sets per wave up to num_perm 128), ("minhash.ties", 0 auto: the second launch tries the
 * tie-tolerant sieve before the dedup pass, 1 = dedup pass only), ("blocks_per_cu", n), ("minhash.prefetch", 0/1/2: never / auto / always),
 * ("minhash.alias", profiling only: >= 0 makes set i read the tokens of set i & mask),
 * ("minhash.p3", 0 auto: three permutations per lane where that walks the fewest slots -- num_perm 129..192, 257..384, 513..576 --, 1 = never three), ("minhash.share", 0 auto: with three or four
 * permutations per lane, lane groups share a last slot that holds at most 32 permutations -- num_perm 129..160, 193..224 --, 1 = off),
 * ("minhash.adapt", 0 auto: the context remembers on the device whether the last call's sets mostly defeated the one-candidate proof and
 * starts the next call with the tie-tolerant one, 1 = off),
 * ("weighted.kernel", 0 auto: dense rows of up to 4096 columns (a multiple of 4) with 65..256 or 321..384 samples through the fetcher / walker kernel, other dense rows
 * of that width through the one-wave-per-row kernel, 1 = the workgroup-per-row kernel, 2 = one wave per row, sample chunks one after the other),
 * ("weighted.refill", 0 auto; 13 = auto without the fetcher / walker split, 5 / 6 / 8 / 9 = the split with other stripe counts and cached list positions,
 * 1 = round 4's plain loads behind the walk, 2 / 3 = the one-wave-per-row kernel's fetch modes; A/B), ("weighted.plan", 1 = plan and tables in two launches), ("weighted.rescue", n: a walk's last n lanes
 * are taken over by the whole wave, 0 auto = 8, < 0 never),
 * ("weighted.path", 0 auto: dense rows through the bound-ordered walk, CSR rows through the row-block kernels,
 * 1 IEEE division for every element, 2 = every element evaluated: dense rows compacted to CSR first),
 * ("weighted.direct", n: a dense row with at most n stored elements per 1000 columns is evaluated element by element
 * instead of walked, default 100), ("weighted.split", 0 auto: the waves of a workgroup that share 64 samples split the list of a
 * dense row that is evaluated entry by entry, 1 = one wave per 64 samples), ("weighted.tail", profiling: 1 .. 5 force the share of a call's logs that
 * counts as "above the cut" to 0.5, 1, 2, 4, 8 %; 0 = the cheapest by the plan's estimate), ("weighted.debug", profiling only: 1 = stage and scan the rows without walking them,
 * 2 = skip the scan, 4 = (fetcher / walker kernel) every stripe keeps its first row: the walkers alone; results are meaningless;
 * 8 = (fetcher / walker kernel, results unaffected) a hand-over wait that outlasts 2^24 polls traps instead of waiting on),
 * ("host.chunk_bytes", see
 * mhx_minhash_bulk), ("lsh.sort_bits", bits of (band, digest) mhx_lsh_sort_bands hands to the radix sort,
 * 0 = chosen from n; the order is exact for any value, fewer bits leave more to the clean-up pass),
 * ("lsh.gather", 1 = gather the full digests after the sort instead of letting them ride through it),
 * ("lsh.prehash", mhx_lsh_sort_bands on a signature matrix: 0 auto = band digests first (one pass at the stream's rate), then the bucketing; 1 = hash inside
 * the bucketing's first pass),
 * ("lsh.team", bucketing over unit-stride sources: 0 auto = one team of 1024 threads x 8 rows per workgroup from eight items per CU on,
 * 256 = teams of 256 threads x 16 rows (until round 6), 1024 = the one team at any size),
 * ("lsh.bigbins", 0 auto: between 2.56M and 10.2M rows one scatter level into 1024 bins of up to 11 264 elements and the bin pass's big form
 * (two passes over the keys instead of three), 1 = never, 2 = from 4 bins on (tests)),
 * ("weighted.min_dim", dense rows at least this wide -- a multiple of 4, up to 4096 columns -- go to the one-wave-per-row / fetcher-walker kernels:
 * 0 auto = 4, 1024 = the rule until round 6),
 * ("lsh.sort", 0 auto: mhx_lsh_sort_bands buckets the bands in two passes (three beyond 10.2M rows) and falls back to the radix sort when a bin
 * overflows or n > 41M, 1 = radix sort always). */
MHX_API int mhx_ctx_set_option(mhx_ctx *ctx, const char *key, int64_t value);
/* Kernel event counters since the last call (synchronises the stream, then resets them):
 *   out[0] sets the sieve launch left to the full launch (failed proof, or skipped by the back-off),
 *   out[1] sets redone with the exact fold, out[2] 256-token sieve blocks evaluated,
 *   out[3] sets the full launch had to hash pair by pair (its dedup sieve failed too).
 * enable != 0 starts/keeps counting, 0 stops it (counting costs one atomic per event). */
#define MHX_NUM_COUNTERS 4
MHX_API int mhx_ctx_counters(mhx_ctx *ctx, int enable, uint64_t out[MHX_NUM_COUNTERS]);
/* What the previous MinHash call on this context learned about the corpus and the first launch of the next call acts on:
 * 0 = the one-candidate proof goes first, 1 = most sets defeated it (the tie-tolerant proof goes first), 2 = heavily
 * repeated tokens.  Timings depend on it (so a benchmark may want to know, or to reset it), results never.  reset != 0 puts it
 * back to 0, the state of a fresh context.  mode may be NULL.  Blocking. */
MHX_API int mhx_ctx_minhash_mode(mhx_ctx *ctx, int reset, int *mode);
/* Diagnostics for parity audits: which sets of the LAST mhx_minhash_bulk_dev call on this context left the fast path.  The
 * result of every set is exact whatever its flag says (ref: datasketch/minhash.py:293-297) -- the flag names the launch that
 * produced it, so that a test can check exactly the sets the rare-event paths handled: flags[i] = 0 the first launch's proof
 * held for set i, 1 = set i was done again by the second launch (tie-tolerant / dedup sieve), 2 = hashed pair by pair by the
 * third.  n_sets must be the last call's; a call that kept no flags (one huge set split over waves, minhash.path != 0) is
 * MHX_ERR_INVALID.  With more than 256 permutations (several passes over the sets) the flags are the last pass's.  Blocking. */
MHX_API int mhx_ctx_minhash_flags(mhx_ctx *ctx, int64_t n_sets, uint8_t *flags);

/* ---- device memory + events (so callers can keep corpora resident and time kernels) ------- */
MHX_API int mhx_dev_alloc(mhx_ctx *ctx, size_t bytes, void **dptr);
MHX_API int mhx_dev_free(mhx_ctx *ctx, void *dptr);
/* Debugging: guard pages.  From this call on every device allocation of the library in this process (mhx_dev_alloc, the
 * contexts' staging buffers, permutations, generator tables) is mapped with the HIP virtual-memory API between two
 * unmapped granules, its LAST byte (align > 0; the size is rounded up to `align` bytes: 16, 8, 4, 1 ...) or its FIRST
 * byte (align < 0) abutting the unmapped range, so that an over- or under-read by a kernel raises a GPU memory access
 * fault instead of touching a neighbour.  align = 0 switches back to hipMalloc (live guarded blocks stay valid).  The
 * environment variable MHX_GUARD_ALLOC=<align> does the same from the first allocation (a value that is not 0 or
 * +-(a power of two <= 4096) is reported on stderr and ignored).  granule (may be NULL) receives the mapping granularity
 * in bytes, live (may be NULL) the number of guarded blocks currently allocated.  A debugging mode, not a production one:
 * the virtual address ranges of freed guarded blocks are never reused (a stale pointer must fault, not alias), so a long
 * guarded run grows its address space by (size + 2 granules) per allocation, and every free synchronises the device. */
MHX_API int mhx_debug_guard_alloc(int align, int64_t *granule, int64_t *live);
/* Debugging: poison.  From this call on every fresh device allocation of the library in this process is filled with
 * byte_value (0..255; -1 switches it off) before it is handed out -- what a board that has been in use gives a process
 * anyway, made deterministic: a kernel that reads a word nobody wrote then computes with 0xFF..FF (a NaN, a -1, an
 * offset of 2^64-1) on every box instead of with the zeros of a freshly booted one.  Environment: MHX_POISON_ALLOC=<byte>.
 * (The library's blocks inside guard mode keep hipMalloc's 256-byte alignment; only mhx_dev_alloc's blocks are placed
 * with the guard alignment as given.) */
MHX_API int mhx_debug_poison_alloc(int byte_value);
/* page-locked host memory: buffers a caller fills and hands to the host entry points again and again (the pieces of
 * mhx_weighted_dense_feed, staging for mhx_minhash_bulk) go up by DMA straight from it, about 1.3x the rate of pageable memory */
MHX_API int mhx_host_alloc(mhx_ctx *ctx, size_t bytes, void **ptr);
/* ctx may be NULL once the allocating context has been destroyed (mhx_ctx_destroy does not free these blocks) */
MHX_API int mhx_host_free(mhx_ctx *ctx, void *ptr);
MHX_API int mhx_memcpy_h2d(mhx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
MHX_API int mhx_memcpy_d2h(mhx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* device-to-device copy, enqueued on the context's stream (no synchronisation) */
MHX_API int mhx_memcpy_d2d(mhx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);
MHX_API int mhx_memset_dev(mhx_ctx *ctx, void *dst_dev, int byte_value, size_t bytes);

MHX_API int mhx_event_create(mhx_ctx *ctx, mhx_event **ev);
MHX_API int mhx_event_record(mhx_event *ev); /* on the owning context's stream */
MHX_API int mhx_event_synchronize(mhx_event *ev);
MHX_API int mhx_event_elapsed_ms(mhx_event *start, mhx_event *stop, float *ms);
MHX_API int mhx_event_destroy(mhx_event *ev);

/* ---- MinHash ------------------------------------------------------------------------------ */
/*
 * Upload permutations (a[k], b[k]) = MinHash.permutations.
 * Replaces ref: datasketch/minhash.py:160-165 (_ensure_gpu_caches).  a, b are host arrays of
 * num_perm uint64 as produced by ref: datasketch/minhash.py:170-184 (kept on the host: legacy
 * numpy RandomState stream).  Any uint64 values are accepted (the reference draws a in [1,p),
 * b in [0,p) with p = 2^61-1, but user-supplied permutations are not range-checked there either).
 */
MHX_API int mhx_perm_create(mhx_ctx *ctx, const uint64_t *a, const uint64_t *b, int32_t num_perm,
                            mhx_perm **perm);
MHX_API int mhx_perm_destroy(mhx_perm *perm);

/*
 * Bulk MinHash over a corpus of pre-hashed token sets, device-resident.
 * Replaces the per-set loop of ref: datasketch/minhash.py:491-522 (generator / bulk) around
 * ref: datasketch/minhash.py:293-297 (update_batch body):
 *     out[i,k] = min( init[i,k],  min_t ((hv[t]*a[k] + b[k]) mod 2^64) mod (2^61-1) & 0xFFFFFFFF )
 * Bit-exact with the numpy path, including the uint64 wrap-around of hv*a+b.
 *
 *   d_hv        token hash values, hv_dtype = MHX_U64 (any uint64) or MHX_U32
 *   d_offsets   int64[n_sets+1] CSR row pointers into d_hv, or NULL for fixed-length sets
 *   fixed_len   tokens per set when d_offsets == NULL (ignored otherwise)
 *   total_tokens  number of elements in d_hv (offsets[n_sets] or n_sets*fixed_len)
 *   d_init      NULL (fresh state 2^32-1, ref :167-168), or uint64 state with row stride
 *               init_stride elements (0 = one [K] prototype shared by every set, K = [n,K])
 *   d_out       [n_sets, K] of out_dtype (MHX_U64 = reference layout; MHX_U32 = compact:
 *               values are < 2^32 unless an init value >= 2^32 survives an empty set, which
 *               MHX_U32 output rejects by saturating to 2^32-1)
 * An empty set leaves its state untouched (ref :265-266).
 */
MHX_API int mhx_minhash_bulk_dev(mhx_perm *perm, const void *d_hv, int hv_dtype,
                                 const int64_t *d_offsets, int64_t fixed_len, int64_t n_sets,
                                 int64_t total_tokens, const uint64_t *d_init, int64_t init_stride,
                                 void *d_out, int out_dtype);

/* Same computation from/to host buffers (H2D, kernel, D2H; blocking).  offsets may be NULL
 * with fixed_len, init may be NULL.  out: uint64 [n_sets, K].  Corpora above 256 MiB are cut
 * into pieces of whole sets whose upload, kernels and download overlap (option
 * "host.chunk_bytes": > 0 piece size in bytes, 0 automatic, < 0 one piece). */
MHX_API int mhx_minhash_bulk(mhx_perm *perm, const uint64_t *hv, const int64_t *offsets,
                             int64_t fixed_len, int64_t n_sets, const uint64_t *init,
                             int64_t init_stride, uint64_t *out);

/* The same with the element types of mhx_minhash_bulk_dev: hv of hv_dtype (MHX_U32: tokens in the range of
 * sha1_hash32, ref: datasketch/hashfunc.py:5-15 -- half the bytes over PCIe), out of out_dtype (MHX_U32: values
 * are < 2^32 by construction, ref: minhash.py:31,297).  mhx_minhash_bulk is the (MHX_U64, MHX_U64) case. */
MHX_API int mhx_minhash_bulk_typed(mhx_perm *perm, const void *hv, int hv_dtype, const int64_t *offsets,
                                   int64_t fixed_len, int64_t n_sets, const uint64_t *init,
                                   int64_t init_stride, void *out, int out_dtype);

/*
 * The reference's default token hash on the device: out[i] = sha1_hash32(token i) (MHX_U32,
 * ref: datasketch/hashfunc.py:5-15: first 4 bytes of the SHA-1 digest, little-endian) or
 * sha1_hash64 (MHX_U64, ref: hashfunc.py:17-28), replacing the per-token Python call of
 * ref: datasketch/minhash.py:221,262-263.  Tokens are byte strings packed back to back in `bytes`;
 * token i is bytes[byte_offsets[i] .. byte_offsets[i+1]) (int64[n_tokens+1], byte_offsets[0] == 0).
 * The uint32 output can be passed straight to mhx_minhash_bulk_dev as hv with hv_dtype MHX_U32.
 */
MHX_API int mhx_sha1_tokens_dev(mhx_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_byte_offsets,
                                int64_t n_tokens, int out_dtype, void *d_out);
MHX_API int mhx_sha1_tokens(mhx_ctx *ctx, const uint8_t *bytes, const int64_t *byte_offsets,
                            int64_t n_tokens, int out_dtype, void *out);
/*
 * MinHash.bulk / generator on raw byte tokens with the default hashfunc (ref: minhash.py:491-522
 * with hashfunc = sha1_hash32): H2D of the packed bytes, SHA-1 kernel, MinHash kernel, D2H of
 * out[n_sets, K] uint64.  set_offsets int64[n_sets+1] indexes TOKENS (set i owns tokens
 * set_offsets[i] .. set_offsets[i+1]); init as in mhx_minhash_bulk.
 */
MHX_API int mhx_minhash_bulk_bytes(mhx_perm *perm, const uint8_t *bytes, const int64_t *byte_offsets,
                                   int64_t n_tokens, const int64_t *set_offsets, int64_t n_sets,
                                   const uint64_t *init, int64_t init_stride, uint64_t *out);

/* hash_dtype = MHX_U32: sha1_hash32 (mhx_minhash_bulk_bytes), MHX_U64: sha1_hash64 (ref: hashfunc.py:17-28). */
MHX_API int mhx_minhash_bulk_bytes_typed(mhx_perm *perm, const uint8_t *bytes, const int64_t *byte_offsets,
                                         int64_t n_tokens, int hash_dtype, const int64_t *set_offsets,
                                         int64_t n_sets, const uint64_t *init, int64_t init_stride,
                                         uint64_t *out);

/*
 * One update_batch on one MinHash state (the reference's GPU seam itself,
 * ref: datasketch/minhash.py:281-291): hashvalues[K] is read, min-combined with the
 * permuted minima of hv[0..n), and written back.  n == 0 is a no-op.
 */
MHX_API int mhx_minhash_update_batch(mhx_perm *perm, const uint64_t *hv, int64_t n,
                                     uint64_t *hashvalues);

/* Elementwise min of two signature matrices (ref: datasketch/minhash.py:337-359 merge,
 * :411-462 union, lean_minhash.py:237-253), count = rows*K elements.  d_out may alias d_x. */
MHX_API int mhx_minhash_merge_dev(mhx_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y,
                                  int64_t count, uint64_t *d_out);
MHX_API int mhx_minhash_merge(mhx_ctx *ctx, const uint64_t *x, const uint64_t *y, int64_t count,
                              uint64_t *out);

/* ---- Weighted MinHash --------------------------------------------------------------------- */
/*
 * Upload generator parameters rs, ln_cs, betas: float32 [sample_size, dim] row-major, exactly the
 * arrays of ref: datasketch/weighted_minhash.py:119-121 (drawn on the host by numpy).
 */
MHX_API int mhx_wgen_create(mhx_ctx *ctx, const float *rs, const float *ln_cs, const float *betas,
                            int32_t sample_size, int32_t dim, mhx_wgen **gen);
MHX_API int mhx_wgen_destroy(mhx_wgen *gen);

/*
 * WeightedMinHashGenerator.minhash_many over a CSR matrix (ref: datasketch/weighted_minhash.py:161-247).
 *   indptr int64[n_rows+1], indices int32[nnz] (sorted within a row, ref :193), values float32[nnz]
 *   values_are_logs != 0: values already hold ln(x) as float32 (parity mode: the host computes
 *     np.log with the same numpy the reference uses; everything downstream is IEEE float32 with
 *     no FMA contraction, so (k, t) are bit-exact);  == 0: the device takes logf(x) itself.
 *   out int64[n_rows, sample_size, 2] = (k, t) pairs (ref :233-239); rows without stored values
 *   get nonempty[row] = 0 (the reference returns None for them, ref :242-247) and zeros in out.
 *   The host entry checks indptr (monotone) and every column index (0 <= index < dim; the reference raises
 *   IndexError there) before anything is uploaded; the _dev entry takes resident arrays as they are: column
 *   indices outside [0, dim) are the caller's error and are not looked for on the device.
 */
MHX_API int mhx_weighted_minhash_many(mhx_wgen *gen, const int64_t *indptr, const int32_t *indices,
                                      const float *values, int values_are_logs, int64_t n_rows,
                                      int64_t *out, uint8_t *nonempty);
MHX_API int mhx_weighted_minhash_many_dev(mhx_wgen *gen, const int64_t *d_indptr,
                                          const int32_t *d_indices, const float *d_values,
                                          int values_are_logs, int64_t n_rows, int64_t nnz,
                                          int64_t *d_out, uint8_t *d_nonempty);

/* The same for DENSE rows x[n_rows, dim] float32 (what the reference first converts with scipy on the host):
 * an entry counts as stored iff its value is not 0 (NaN counts, as for scipy's nonzero()); with
 * values_are_logs != 0 the caller passes ln(x) and ln(0) = -inf marks the absent entries.  The CSR form is
 * built on the device. */
MHX_API int mhx_weighted_minhash_many_dense(mhx_wgen *gen, const float *x, int values_are_logs,
                                            int64_t n_rows, int64_t *out, uint8_t *nonempty);
MHX_API int mhx_weighted_minhash_many_dense_dev(mhx_wgen *gen, const float *d_x, int values_are_logs,
                                                int64_t n_rows, int64_t *d_out, uint8_t *d_nonempty);

/* Dense rows arriving in pieces (a host that takes np.log piece by piece ahead of the device -- parity mode of
 * ref: weighted_minhash.py:212 -- or reads the matrix from disk):
 *   begin  sizes the device buffers for pieces of up to piece_rows rows (two of them, so that piece i+1 goes up while
 *          piece i is evaluated);
 *   feed   uploads x[n_rows, dim] and returns as soon as x may be overwritten; the evaluation of this piece is queued
 *          behind it and the results of the PREVIOUS piece are brought down meanwhile.  out / nonempty of a piece
 *          (same layout as above, n_rows rows) are complete when the next feed, or end, has returned;
 *   end    brings down the last piece and releases the buffers (also to be called after an error).
 * One feed at a time per generator's context; other calls on the context may come in between. */
typedef struct mhx_wfeed mhx_wfeed;
MHX_API int mhx_weighted_dense_begin(mhx_wgen *gen, int values_are_logs, int64_t piece_rows, mhx_wfeed **feed);
MHX_API int mhx_weighted_dense_feed(mhx_wfeed *feed, const float *x, int64_t n_rows, int64_t *out, uint8_t *nonempty);
MHX_API int mhx_weighted_dense_end(mhx_wfeed *feed);

/* out[i] = the float32 logarithm the device-log mode (values_are_logs == 0) takes of x[i]: the one place where the
 * fast mode can differ from numpy's float32 log (ref: weighted_minhash.py:212), exposed so that callers can check
 * the difference against their tolerance (BASELINE: 1e-6 relative). */
MHX_API int mhx_weighted_logf(mhx_ctx *ctx, const float *x, int64_t n, float *out);

/* ---- Packing for downstream consumers ----------------------------------------------------- */
/* Number of uint64 blocks bBitMinHash uses for num_perm values of b bits
 * (ref: datasketch/b_bit_minhash.py:147-172). */
MHX_API int mhx_bbit_num_blocks(int32_t num_perm, int32_t b, int32_t *num_blocks);
/* b-bit packing of a whole signature matrix in the bit order of ref:
 * datasketch/b_bit_minhash.py:37-38,82-97.  sig [n,K] uint64 -> out [n,num_blocks] uint64. */
MHX_API int mhx_bbit_pack_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n_sigs, int32_t num_perm,
                              int32_t b, uint64_t *d_out);
MHX_API int mhx_bbit_pack(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                          int32_t b, uint64_t *out);
/* MinHashLSH band keys: out[n, bands*r] holds each hashvalue byte-swapped to big-endian so that
 * bytes(out[i, j*r:(j+1)*r]) is the key of band j (ref: datasketch/lsh.py:199,344,537-538). */
MHX_API int mhx_band_keys_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n_sigs, int32_t num_perm,
                              int32_t bands, int32_t r, uint64_t *d_out);
MHX_API int mhx_band_keys(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                          int32_t bands, int32_t r, uint64_t *out);
/* 64-bit band digests for device-side bucketing: out[n, bands], out[i,j] = FNV-1a-64 of the band key
 * bytes of band j of row i (the very bytes of mhx_band_keys; what ref: datasketch/lsh.py:540-543 stores
 * for MinHashLSH(hashfunc=fnv1a_64)).  Equal digests <=> same LSH bucket (up to 2^-64 collisions). */
MHX_API int mhx_band_digests_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n_sigs, int32_t num_perm,
                                 int32_t bands, int32_t r, uint64_t *d_out);
MHX_API int mhx_band_digests(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                             int32_t bands, int32_t r, uint64_t *out);
/* LSH bucketing by sort (what the per-band dictionaries of ref: datasketch/lsh.py:326-347,370-400 do by
 * hashing): for every band j, sorted_digests[j*n .. (j+1)*n) are the band-j digests of all n signatures in
 * ascending order and sorted_rows[...] the row numbers in the same order -- every LSH bucket of band j is
 * a run of equal digests.  Rows are uint32 (n < 2^32).  The default two-pass bucketing reads one flag back between its
 * passes (the call synchronises the stream once; with option "lsh.sort" = 1, the radix sort, it only enqueues) and keeps
 * its bin slabs -- about 30 bytes per (row, band) -- in the context's scratch until mhx_ctx_release_scratch. */
MHX_API int mhx_lsh_sort_bands_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n_sigs, int32_t num_perm,
                                   int32_t bands, int32_t r, uint64_t *d_sorted_digests,
                                   uint32_t *d_sorted_rows);
MHX_API int mhx_lsh_sort_bands(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                               int32_t bands, int32_t r, uint64_t *sorted_digests, uint32_t *sorted_rows);
/* The same from digests that are already there: d_digests [n_sigs, bands] as mhx_band_digests* wrote them (a caller that
 * keeps the digest matrix -- config 3 stores it as the index's keys -- does not pay for hashing the bands twice). */
MHX_API int mhx_lsh_sort_digests_dev(mhx_ctx *ctx, const uint64_t *d_digests, int64_t n_sigs, int32_t bands,
                                     uint64_t *d_sorted_digests, uint32_t *d_sorted_rows);
/* Candidate pairs: every pair of rows i < j that share the digest of at least one band -- the rows
 * MinHashLSH.query (ref: datasketch/lsh.py:370-400) would return for each other -- from the output of
 * mhx_lsh_sort_bands.  pairs: int64[capacity, 2], ascending by (i, j), unique.  *n_pairs receives the
 * number of unique pairs; when it exceeds capacity nothing is written and the caller calls again with
 * a larger buffer.  *n_raw (may be NULL) receives the pair count before deduplication across bands.
 * A bucket of L equal keys holds L(L-1)/2 pairs: MHX_ERR_OOM when they do not fit in device memory.
 * Blocking (the counts come back to the host). */
MHX_API int mhx_lsh_candidate_pairs_dev(mhx_ctx *ctx, const uint64_t *d_sorted_digests,
                                        const uint32_t *d_sorted_rows, int64_t n_sigs, int32_t bands,
                                        int64_t *d_pairs, int64_t capacity, int64_t *n_pairs,
                                        int64_t *n_raw);
/* signatures (host) -> band digests -> per-band sort -> candidate pairs (host), one call */
MHX_API int mhx_lsh_candidate_pairs(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                                    int32_t bands, int32_t r, int64_t *pairs, int64_t capacity,
                                    int64_t *n_pairs, int64_t *n_raw);
/* Bulk query: the keys MinHashLSH.query (ref: datasketch/lsh.py:370-431) would return for each of m probe
 * signatures against an index of n signatures held as sorted bands (mhx_lsh_sort_bands*): for every band the
 * probe's digest is located by binary search and the matching run is its bucket.  pairs: int64[capacity, 2] =
 * (probe, index row), ascending, unique; *n_pairs as in mhx_lsh_candidate_pairs_dev (larger than capacity: nothing
 * written, call again).  d_index_sig (may be NULL): the index's own [n, num_perm] matrix of the same sig_dtype;
 * when given, a candidate is kept only if the r words of the band really are equal, so a 64-bit digest
 * collision between different band keys cannot produce a row the reference's dictionaries would not.  Blocking. */
MHX_API int mhx_lsh_query_dev(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows,
                              int64_t n_sigs, int32_t bands, int32_t r, const void *d_query_sig,
                              const void *d_index_sig, int sig_dtype, int32_t num_perm, int64_t n_queries,
                              int64_t *d_pairs, int64_t capacity, int64_t *n_pairs);
/* Batched MinHash.jaccard numerators (ref: datasketch/minhash.py:299-324): counts[p] = number of equal
 * positions of rows pairs[p][0] of sig_a and pairs[p][1] of sig_b (both [*, num_perm] uint64; may be the
 * same matrix); the estimate is counts / num_perm.  pairs int64[n_pairs, 2]. */
MHX_API int mhx_jaccard_pairs_dev(mhx_ctx *ctx, const uint64_t *d_sig_a, const uint64_t *d_sig_b,
                                  int32_t num_perm, const int64_t *d_pairs, int64_t n_pairs,
                                  int32_t *d_counts);
MHX_API int mhx_jaccard_pairs(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                              const int64_t *pairs, int64_t n_pairs, int32_t *counts);
/* Batched bBitMinHash.jaccard numerators (ref: datasketch/b_bit_minhash.py:53-72) on packed rows (mhx_bbit_pack*):
 * counts[p] = number of the num_perm positions whose b-bit values agree in rows pairs[p][0] of blocks_a and
 * pairs[p][1] of blocks_b (both [*, num_blocks] uint64; may be the same matrix) -- XOR + popcount on the packed
 * blocks, nothing is unpacked.  The estimate is (counts / num_perm - C1) / (1 - C2) with the reference's C1, C2. */
MHX_API int mhx_bbit_jaccard_pairs_dev(mhx_ctx *ctx, const uint64_t *d_blocks_a, const uint64_t *d_blocks_b,
                                       int32_t num_perm, int32_t b, const int64_t *d_pairs, int64_t n_pairs,
                                       int32_t *d_counts);
MHX_API int mhx_bbit_jaccard_pairs(mhx_ctx *ctx, const uint64_t *blocks, int64_t n_rows, int32_t num_perm, int32_t b,
                                   const int64_t *pairs, int64_t n_pairs, int32_t *counts);
/* The device entry points above for signature matrices of sig_dtype MHX_U64 or MHX_U32 -- uint32 is the compact
 * output of mhx_minhash_bulk_dev and the wire format of the all-gather (values are < 2^32, ref: minhash.py:31,297);
 * results are those of the widened matrix. */
MHX_API int mhx_bbit_pack_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs,
                                    int32_t num_perm, int32_t b, uint64_t *d_out);
MHX_API int mhx_band_digests_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs,
                                       int32_t num_perm, int32_t bands, int32_t r, uint64_t *d_out);
/* Config 5 in one call: the b-bit blocks (mhx_bbit_pack*: ref datasketch/b_bit_minhash.py:78-101) AND the band digests
 * (mhx_band_digests*: ref datasketch/lsh.py:199,344,537-543) of the same [n, num_perm] matrix.  When bands is a power of
 * two <= 64, r is 4, 8 or 16, bands * r == num_perm and rows are 16-byte aligned, ONE kernel reads the matrix once and
 * writes both outputs (*fused = 1); any other shape runs the two kernels one after the other (*fused = 0).  Results are
 * those of the two separate calls, bit for bit.  d_blocks: uint64[n, num_blocks], d_digests: uint64[n, bands] or, with digest_layout =
 * MHX_BAND_MAJOR, uint64[bands, n]. */
MHX_API int mhx_bbit_pack_band_digests_dev(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs,
                                           int32_t num_perm, int32_t b, int32_t bands, int32_t r, int digest_layout,
                                           uint64_t *d_blocks, uint64_t *d_digests, int *fused);
/* Layout of a band-digest matrix on the device.  MHX_ROW_MAJOR: [n, bands] (row i's digests together: the keys of one
 * signature, what a host-side index wants).  MHX_BAND_MAJOR: [bands, n] (band j's digests of all rows together: one
 * array per hashtable, ref datasketch/lsh.py:199 -- and what the bucketing reads with unit stride: from a row-major
 * matrix every 128-byte line is fetched by the four XCDs whose bands share it, 1.30 GB of reads for a 320 MB matrix). */
#define MHX_ROW_MAJOR 0
#define MHX_BAND_MAJOR 1
MHX_API int mhx_band_digests_layout_dev(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs,
                                        int32_t num_perm, int32_t bands, int32_t r, int layout, uint64_t *d_out);
MHX_API int mhx_lsh_sort_digests_layout_dev(mhx_ctx *ctx, const uint64_t *d_digests, int64_t n_sigs, int32_t bands,
                                            int layout, uint64_t *d_sorted_digests, uint32_t *d_sorted_rows);
MHX_API int mhx_lsh_sort_bands_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs,
                                         int32_t num_perm, int32_t bands, int32_t r,
                                         uint64_t *d_sorted_digests, uint32_t *d_sorted_rows);
MHX_API int mhx_jaccard_pairs_dev_typed(mhx_ctx *ctx, const void *d_sig_a, const void *d_sig_b, int sig_dtype,
                                        int32_t num_perm, const int64_t *d_pairs, int64_t n_pairs,
                                        int32_t *d_counts);
/* LeanMinHash.serialize of every row, little-endian: n records of 12+4*K bytes
 * (ref: datasketch/lean_minhash.py:126-175). */
MHX_API int mhx_lean_serialize_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n_sigs,
                                   int32_t num_perm, int64_t seed, uint8_t *d_out);
MHX_API int mhx_lean_serialize(mhx_ctx *ctx, const uint64_t *sig, int64_t n_sigs, int32_t num_perm,
                               int64_t seed, uint8_t *out);
/* The same for a matrix of sig_dtype (MHX_U32: the compact form) in either byte order the reference's `byteorder` argument can ask
 * for: MHX_LITTLE_ENDIAN is '<' (and '@' / '=' on this little-endian host), MHX_BIG_ENDIAN is '>' / '!' (network order). */
#define MHX_LITTLE_ENDIAN 0
#define MHX_BIG_ENDIAN 1
MHX_API int mhx_lean_serialize_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n_sigs, int32_t num_perm,
                                         int64_t seed, int byteorder, uint8_t *d_out);
/* The inverse: LeanMinHash.deserialize of n records of 12 + 4*K bytes laid back to back (ref: datasketch/lean_minhash.py:177-214):
 * d_sig [n, K] of sig_dtype receives the hash values, d_seeds (may be NULL) int64[n] each record's seed.  A record whose length
 * field is not num_perm is counted in *d_bad (uint32 on the device, may be NULL; the caller zeroes it) -- the host entry point
 * returns MHX_ERR_INVALID for such a buffer.  Records need 4-byte alignment. */
MHX_API int mhx_lean_deserialize_dev(mhx_ctx *ctx, const uint8_t *d_records, int64_t n_sigs, int32_t num_perm, int byteorder,
                                     int sig_dtype, void *d_sig, int64_t *d_seeds, uint32_t *d_bad);
MHX_API int mhx_lean_deserialize(mhx_ctx *ctx, const uint8_t *records, int64_t n_sigs, int32_t num_perm, int byteorder,
                                 uint64_t *sig, int64_t *seeds);
/* The inverse of mhx_bbit_pack*: bBitMinHash.__setstate__ of every row (ref: datasketch/b_bit_minhash.py:103-125) --
 * blocks [n, num_blocks] uint64 -> the b-bit values [n, num_perm] uint32 (what a restored bBitMinHash holds as hashvalues). */
MHX_API int mhx_bbit_unpack_dev(mhx_ctx *ctx, const uint64_t *d_blocks, int64_t n_sigs, int32_t num_perm, int32_t b,
                                uint32_t *d_out);
MHX_API int mhx_bbit_unpack(mhx_ctx *ctx, const uint64_t *blocks, int64_t n_sigs, int32_t num_perm, int32_t b, uint32_t *out);

/* ---- Multi-GPU: assemble the signature matrix (RCCL over xGMI) ----------------------------- */
/* 128-byte RCCL unique id, created on rank 0 and distributed by the caller (env, file, socket). */
#define MHX_COMM_ID_BYTES 128
/* RCCL (librccl.so) is bound with dlopen at the first mhx_comm_* call, so libmhx loads on hosts without it. */
MHX_API int mhx_comm_unique_id(uint8_t id[MHX_COMM_ID_BYTES]);
MHX_API int mhx_comm_create(mhx_ctx *ctx, const uint8_t id[MHX_COMM_ID_BYTES], int rank,
                            int world_size, mhx_comm **comm);
MHX_API int mhx_comm_destroy(mhx_comm *comm);
/* What RCCL itself reports for the communicator: ncclCommUserRank, ncclCommCount ("ranks seen"),
 * ncclCommCuDevice and ncclGetVersion.  Any out pointer may be NULL. */
MHX_API int mhx_comm_info(mhx_comm *comm, int *rank, int *world_size, int *device, int *rccl_version);
/* All-gather equal-sized row shards: every rank contributes bytes_per_rank bytes from d_send,
 * d_recv receives world_size*bytes_per_rank bytes in rank order.  Enqueued on the ctx stream. */
MHX_API int mhx_comm_allgather_dev(mhx_comm *comm, const void *d_send, void *d_recv,
                                   size_t bytes_per_rank);
/* All-gather of UNEQUAL row shards, in place: rank q contributes recv_bytes[q] bytes (its own entry from d_send),
 * and every rank receives them at d_recv + recv_offsets[q].  recv_offsets / recv_bytes: host arrays of world_size
 * entries, identical on every rank (shard sizes are common knowledge: SURVEY.md section 8e "AllGatherv-style with
 * per-rank counts").  One grouped launch of ncclBroadcast per root; enqueued on the ctx stream, no host sync. */
MHX_API int mhx_comm_allgatherv_dev(mhx_comm *comm, const void *d_send, void *d_recv,
                                    const uint64_t *recv_offsets, const uint64_t *recv_bytes);
/* A grouped point-to-point exchange (ncclSend / ncclRecv inside one group call, one launch): the BY-BAND exchange of band
 * digests.  The reference's index is one independent hashtable per band (ref: datasketch/lsh.py:199,326-347), so the rank that
 * builds the tables of bands [lo, hi) needs those bands' digests of ALL rows -- [hi - lo, N] uint64 -- and nothing else: every
 * rank digests its own rows band-major ([bands, n_p]) and sends each peer the runs of that peer's bands (at 10M rows x 32 bands
 * on 8 ranks: 0.28 GB received per GPU where the all-gather of the uint32 signature matrix receives 8.96 GB).
 *   message i of the send list: send_bytes[i] bytes at d_send + send_offsets[i] go to rank send_peers[i];
 *   message i of the receive list: recv_bytes[i] bytes from rank recv_peers[i] land at d_recv + recv_offsets[i].
 * Between a pair of ranks messages are matched in list order.  A message a rank sends to itself is a device copy on the stream
 * (its k-th such send pairs with its k-th such receive; sizes must agree).  Messages of zero bytes are dropped on both sides.
 * All lists are host arrays; enqueued on the ctx stream, no host synchronisation.  d_send and d_recv must not overlap. */
MHX_API int mhx_comm_exchange_dev(mhx_comm *comm, const void *d_send, void *d_recv, int32_t n_send, const int32_t *send_peers,
                                  const uint64_t *send_offsets, const uint64_t *send_bytes, int32_t n_recv,
                                  const int32_t *recv_peers, const uint64_t *recv_offsets, const uint64_t *recv_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MHX_H_ */

// lsh_kernels.hip -- device-side LSH bucketing.
//
// Reference: MinHashLSH keeps one dictionary per band, keyed by the band's key bytes
// (datasketch/lsh.py:326-347 insert, :370-400 query); two signatures are candidates iff they share a
// key in at least one band.  Here the grouping is a sort: per band, the 64-bit digests of the band keys
// (pack_kernels.hip: FNV-1a-64 of exactly the reference's key bytes) are sorted together with the row
// numbers, so every bucket becomes a run of equal digests.  The bucketing itself is hand-written (two or three passes, below);
// its fallback for corpora it cannot bin, and the sort of the raw candidate pairs, are rocPRIM's device radix sort (a library
// primitive) -- everything around it, including the scans and the deduplication, is ours.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "band_digest.h"
#include "mhx_internal.h"

namespace mhx {
namespace {

// All bands of all rows go through ONE radix sort (32 separate sorts of 10^6 keys are launch-bound: ~20
// small kernels each).  The sort key is the band and a prefix of the digest (40 bits in all at n = 10^6:
// five 8-bit passes over 12-byte pairs) with the row as the value; sorting 64 + band_bits bits of a 16-byte
// key took nine passes: 3.5 ms instead of 2.5 for 32 x 10^6 keys.  Exactness does not rest on the prefix:
// after the sort the full digests are gathered in order and every run of equal prefixes whose digests are
// not all equal is put in (digest, row) order in place.  Option "lsh.sort_bits" overrides the key length
// (tests use short ones so that mixed runs abound).

// the top sort_bits bits of (band, digest), right-aligned: the radix sort runs over bits [0, sort_bits)
__device__ __host__ __forceinline__ uint64_t prefix_key(uint64_t digest, uint32_t band, int band_bits, int sort_bits) {
    const uint64_t whole = ((uint64_t)band << (64 - band_bits)) | (digest >> band_bits);
    return whole >> (64 - sort_bits);
}

// digests[n, bands] (row-major) -> (key, row) in the same order: rows ascend within every band and the
// sort is stable, so equal keys of a band keep ascending rows
__global__ __launch_bounds__(256) void band_keys_for_sort_kernel(const uint64_t *__restrict__ digests, int64_t n, int32_t bands,
                                                                 int band_bits, int sort_bits, int band_major, uint64_t *__restrict__ keys,
                                                                 uint32_t *__restrict__ rows) {
    const int64_t total = n * (int64_t)bands;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / bands;
        const int64_t band = idx - row * bands;
        keys[idx] = prefix_key(digests[band_major ? band * n + row : idx], (uint32_t)band, band_bits, sort_bits);
        rows[idx] = (uint32_t)row;
    }
}

// position p of the sorted order belongs to band p / n (every band has n entries): fetch its full digest
__global__ __launch_bounds__(256) void gather_digests_kernel(const uint64_t *__restrict__ digests, const uint32_t *__restrict__ rows,
                                                             int64_t n, int32_t bands, int64_t total, int band_major,
                                                             uint64_t *__restrict__ sorted_digests) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x)
        sorted_digests[p] = digests[band_major ? p / n * n + rows[p] : (int64_t)rows[p] * bands + p / n];
}

// ---- the bands bucketed in two passes (round 3) ------------------------------------------------------------
// FNV digests of band keys are uniform, so a most-significant-digit split lands the elements of a band in bins of
// nearly equal size, and a bin small enough for LDS is finished there: two passes over 12-byte (digest, row) pairs
// instead of the library sort's four over key + value (with the digests' lower bits riding along).
//   pass 1 (lsh_bin_scatter_kernel): a team of 256 threads takes 2048 rows of ONE band, computes their digests (the band's
//     r values are 32 or 64 contiguous bytes of a row; the teams of a workgroup take the bands that share those rows'
//     cache lines), counts them per bin in LDS (the top bin_bits of the digest), reserves a range in
//     every bin's slab with one global atomic per (workgroup, bin) and writes (digest, row) there.
//   pass 2 (lsh_bin_sort_kernel): one workgroup per (band, bin) loads the slab (about 2400 elements) into LDS, spreads
//     it over 2^kSubBits = 1024 buckets by the next kSubBits = 10 digest bits (two or three elements per bucket), ranks every element inside its
//     bucket by (digest, row) and writes it to its place in the output: band * n + the sizes of the band's bins before
//     it + the bucket's start + the rank.  The order is (band, digest, row): exactly what the stable radix sort of rows in
//     ascending order gives.
// A bin can hold kBinCap elements; one that would overflow (a corpus of near-identical signatures) raises a flag that the
// host reads after pass 1, and the call falls back to the radix sort.
// Round 6: over unit-stride sources (band-major digests, big bins) pass 1 runs as one team of 1024 threads x 8 rows per workgroup
// (launch_scatter_rows), and between 2.56M and 10.2M rows pass 1 stops at 1024 bins per band which pass 2 finishes in its big form
// (bins of up to kBigBinCap = 11 264 elements in a CU's whole LDS): two passes over the keys where rounds 4-5 made three.
constexpr int kBinCap = 3072;
constexpr int kScatterRowsDefault = 8;   // rows per thread of pass 1 (2048 per team: 26 KB of LDS, six one-team workgroups per CU)
constexpr int kSubBits = 10;  // (11 until round 5: 8 KB less LDS puts three workgroups on a CU instead of two, 0.45 -> 0.35 ms for 40M keys; 9 bits and a fourth workgroup gain nothing)
constexpr int kSortThreads = 512;
constexpr int kMaxBinBits = 14;        // final bins per band (two levels beyond kOneLevelBits)
constexpr int kOneLevelBits = 10;      // at most this many bits in one scatter pass
constexpr int kMaxBinsPerThread = (1 << kOneLevelBits) / 256;

// inclusive prefix sum over the threads of a workgroup: shuffles inside a wave, the wave totals through LDS
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t *tmp4, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, o);
        if (lane >= o) v += up;
    }
    if (lane == 63) tmp4[wave] = v;
    __syncthreads();
    uint32_t add = 0;
    for (int w = 0; w < wave; ++w) add += tmp4[w];
    __syncthreads();
    return v + add;
}

__device__ __forceinline__ uint32_t bin_of(uint64_t digest, int bin_bits) { return bin_bits ? (uint32_t)(digest >> (64 - bin_bits)) : 0u; }

// A workgroup is `band_share` teams of 256 threads; team q takes band group * band_share + q of the same 2048 rows.  The
// bands of a group are the ones whose r values of a row share a 128-byte line: the teams load in lock step, so the line
// comes from HBM once (one workgroup per band left that to the L2 -- whose 4 MB turn over in microseconds under this
// stream: 1.04 ms for the pass instead of 0.5).
//
// Two levels (round 5).  With more than 1024 bins a chunk of 2048 rows leaves one or two elements per bin: 8-byte runs, every
// one a partial write (10M rows = 4096 bins: 22 ms for 320M keys, 0.04 of HBM).  So beyond 2^10 bins the split is done in two
// scatter passes of about the square root each: level 0 spreads a band over 2^hi big bins (slabs of `cap` = n / 2^hi + slack
// elements), level 1 takes every big bin as its source (SlabPairs: (digest, row) pairs instead of a signature matrix) and
// spreads it over its 2^lo final bins -- the same slabs and cursors a one-level split would have filled, so pass 2 does not
// know the difference.  A source "unit" is a band (level 0 / one level) or a (band, big bin); its elements' output bin inside
// the band is the top hi + lo digest bits, of which the low lo bits index the team's histogram.
struct SlabPairs {};  // source of level 1: src_dig / src_row slabs of src_cap elements per unit, src_cursor[unit] of them filled

template <typename SigT, int kScatterRows, int T = 256>  // T: threads of a team (1024 for unit-stride sources since round 6: see launch_scatter_rows)
__global__ __launch_bounds__(1024) void lsh_bin_scatter_kernel(const SigT *__restrict__ sig, int32_t k, int32_t r, int64_t n, int32_t units,
                                                               int hi_bits, int lo_bits, int band_share, uint32_t cap,
                                                               uint32_t *__restrict__ cursor, uint64_t *__restrict__ slab_dig,
                                                               uint32_t *__restrict__ slab_row, uint32_t *__restrict__ overflow,
                                                               const uint64_t *__restrict__ src_dig, const uint32_t *__restrict__ src_row,
                                                               const uint32_t *__restrict__ src_cursor, uint32_t src_cap) {
    constexpr bool kPairs = std::is_same<SigT, SlabPairs>::value;
    using RowT = typename std::conditional<kPairs, uint32_t, uint16_t>::type;  // staged per element: the row itself, or row - row0
    constexpr int kChunk = T * kScatterRows;
    extern __shared__ uint64_t scatter_lds[];  // per team: st_dig[kChunk] | hist[nb] | base[nb] | lstart[nb] | st_row RowT[kChunk] | scan_tmp[16]
    const int nb = 1 << lo_bits, team = threadIdx.x / T, tid = threadIdx.x % T;
    const size_t team_words = kChunk + (3 * (size_t)nb * 4 + kChunk * sizeof(RowT) + 64 + 7) / 8;  // (+ 64: scan_tmp, a word per wave of the team -- sixteen at T = 1024)
    uint64_t *st_dig = scatter_lds + team * team_words;  // the chunk's elements grouped by bin before they go out
    uint32_t *hist = reinterpret_cast<uint32_t *>(st_dig + kChunk), *base = hist + nb, *lstart = base + nb;
    RowT *st_row = reinterpret_cast<RowT *>(lstart + nb);
    uint32_t *scan_tmp = reinterpret_cast<uint32_t *>(st_row + kChunk);
    const int64_t span = kPairs ? (int64_t)src_cap : n;  // a unit's elements are [0, count) with count <= span
    const int64_t chunks = (span + kChunk - 1) / kChunk;
    const int groups = units / band_share;
    const int tot_bits = hi_bits + lo_bits;
    // item = unit group fastest: with 8 groups and a grid that is a multiple of 8, group x is always on XCD x (workgroups go round
    // the XCDs), whose L2 then merges the short runs of its bins into whole lines.  (Measured and dropped in round 5: dealing the
    // band groups that share 128-byte input lines of a ROW-major digest matrix to one XCD -- reads 1.30 GB -> 0.32 GB, but 16
    // bands' slabs per XCD no longer merge: writes 0.62 -> 1.02 GB in 32-byte requests, 447 -> 493 us,
    // profiles/r05_pmc_scatter_work_orders.txt.  The band-major input (Digest64BM) keeps both properties.)
    for (int64_t item = blockIdx.x; item < chunks * groups; item += gridDim.x) {
        const int unit = (int)(item % groups) * band_share + team;
        const int64_t row0 = item / groups * kChunk;
        const int64_t count = kPairs ? (int64_t)min(src_cursor[unit], src_cap) : n;
        if (kPairs && row0 >= count) continue;  // (band_share is 1 for pair sources: workgroup-uniform)
        const int band = kPairs ? unit >> hi_bits : unit;
        const int64_t src0 = kPairs ? (int64_t)unit * src_cap : 0;
        for (int t = tid; t < nb; t += T) hist[t] = 0;
        __syncthreads();
        // the element's bin inside the band (top hi + lo digest bits) and inside this source unit (the low lo bits of that)
        const auto gbin_of = [&](uint64_t d) { return tot_bits ? (uint32_t)(d >> (64 - tot_bits)) : 0u; };
        // (Round 6, measured and dropped: the counting atomic returning the element's rank, so that the staging loop needs no second atomic and
        // the output loop one table word instead of two -- seven LDS operations per element instead of nine, five at random addresses instead
        // of seven.  Same box, interleaved: 0.495 / 0.498 ms against 0.491 / 0.451 for 40M keys.  The LDS bank conflicts the counters show
        // (0.555 of the LDS cycles) are not what the pass waits for.)
        // All of the thread's loads go out before any of them is waited for: a thread behind the unit's end reads the unit's last
        // element again (nothing is predicated), and the histogram's LDS atomics come in a loop of their own.  (Until round 5 the
        // atomic sat next to its load inside `if (row < count)`: the compiler put an s_waitcnt vmcnt(0) between every load and its
        // atomic -- sixteen memory latencies in a row per chunk, which is what the pass's 24 us per chunk were made of.)
        uint64_t dg[kScatterRows];
        uint32_t rw[kPairs ? kScatterRows : 1];
#pragma unroll
        for (int j = 0; j < kScatterRows; ++j) {
            const int64_t row = row0 + j * T + tid;
            const int64_t at = row < count ? row : count - 1;  // (count > 0 here)
            if constexpr (kPairs) {
                dg[j] = src_dig[src0 + at];
                rw[j] = src_row[src0 + at];
            } else {
                dg[j] = band_digest_of<SigT>(sig, at, band, k, r, n);
            }
        }
        uint16_t rank[kScatterRows];  // (T >= 1024 only) the element's rank inside its (team, bin) piece: what the counting atomic returns
#pragma unroll
        for (int j = 0; j < kScatterRows; ++j) {
            if constexpr (T >= 1024) rank[j] = row0 + j * T + tid < count ? (uint16_t)atomicAdd(&hist[gbin_of(dg[j]) & (nb - 1)], 1u) : (uint16_t)0;
            else if (row0 + j * T + tid < count) atomicAdd(&hist[gbin_of(dg[j]) & (nb - 1)], 1u);
        }
        __syncthreads();
        // per bin: a range of its slab (one global atomic) and the start of its elements in the team's staging area
        // (exclusive scan of the counts: thread t owns bins [t * per, t * per + per))
        const int64_t out0 = ((int64_t)band << tot_bits) + (kPairs ? (int64_t)(unit & ((1 << hi_bits) - 1)) << lo_bits : 0);  // the unit's first output bin
        {
            const int per = (nb + T - 1) / T;  // (<= kMaxBinsPerThread)
            uint32_t cnts[kMaxBinsPerThread], bases[kMaxBinsPerThread], sum = 0;
#pragma unroll
            for (int j = 0; j < kMaxBinsPerThread; ++j) {  // the thread's returning atomics go out back to back: one round trip to the L2, not `per`
                const int t = tid * per + j;
                cnts[j] = j < per && t < nb ? hist[t] : 0u;
                bases[j] = cnts[j] ? atomicAdd(&cursor[out0 + t], cnts[j]) : 0u;
                sum += cnts[j];
            }
            const uint32_t incl = block_inclusive_scan(sum, scan_tmp, tid);
            uint32_t at = incl - sum;
#pragma unroll
            for (int j = 0; j < kMaxBinsPerThread; ++j) {
                const int t = tid * per + j;
                if (j < per && t < nb) {
                    if (bases[j] + cnts[j] > cap) *overflow = 1u;
                    base[t] = T >= 1024 ? bases[j] - at : bases[j];  // (T >= 1024: slab position of staged element i of this bin = base[bin] + i, mod 2^32)
                    lstart[t] = at;
                    hist[t] = 0;
                    at += cnts[j];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kScatterRows; ++j) {
            const int64_t row = row0 + j * T + tid;
            if (row < count) {
                const uint32_t bin = gbin_of(dg[j]) & (nb - 1);
                uint32_t lp;
                if constexpr (T >= 1024) lp = lstart[bin] + rank[j];
                else lp = lstart[bin] + atomicAdd(&hist[bin], 1u);
                st_dig[lp] = dg[j];
                if constexpr (kPairs) st_row[lp] = rw[j];
                else st_row[lp] = (uint16_t)(j * T + tid);
            }
        }
        __syncthreads();
        // out: consecutive threads carry consecutive elements of a bin -- every (team, bin) piece is one contiguous run
        const uint32_t total = (uint32_t)min((int64_t)kChunk, count - row0);
        for (uint32_t i = tid; i < total; i += T) {
            const uint64_t d = st_dig[i];
            const uint32_t bin = gbin_of(d) & (nb - 1);
            const uint32_t pos = T >= 1024 ? base[bin] + i : base[bin] + (i - lstart[bin]);
            if (pos < cap) {
                const int64_t at = (out0 + bin) * cap + pos;
                slab_dig[at] = d;
                slab_row[at] = kPairs ? (uint32_t)st_row[i] : (uint32_t)(row0 + st_row[i]);
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ bool pair_less(uint64_t da, uint32_t ra, uint64_t db, uint32_t rb) { return da < db || (da == db && ra < rb); }

// where every bin's elements go in the output: the sizes of the band's bins before it (one workgroup per band; a bin holds at
// most kBinCap elements).  Computed once here: lsh_bin_sort_kernel used to sum the band's cursors in every workgroup -- a loop of
// dependent loads (eight L2 round trips per thread at 4096 bins) and a 512-thread tree reduction, nine barriers per bin.
__global__ __launch_bounds__(256) void lsh_bin_offsets_kernel(const uint32_t *__restrict__ cursor, int bin_bits, uint32_t bin_cap, uint32_t *__restrict__ bin_start) {
    __shared__ uint32_t scan_tmp[4];
    const int nb = 1 << bin_bits, tid = threadIdx.x;
    const uint32_t *cur = cursor + (int64_t)blockIdx.x * nb;
    uint32_t *dst = bin_start + (int64_t)blockIdx.x * nb;
    const int per = (nb + 255) / 256;
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) {
        const int t = tid * per + j;
        if (t < nb) sum += min(cur[t], bin_cap);
    }
    uint32_t at = block_inclusive_scan(sum, scan_tmp, tid) - sum;
    for (int j = 0; j < per; ++j) {
        const int t = tid * per + j;
        if (t < nb) {
            dst[t] = at;
            at += min(cur[t], bin_cap);
        }
    }
}

// Round 6: the pass is a template over (capacity, sub-bucket bits, threads).  <3072, 10, 512> is the pass of rounds 3-5 (bins of ~2 400 elements,
// three workgroups per CU).  <kBigBinCap, 12, 1024> finishes bins of ~10 000 elements -- the whole LDS of a CU, one workgroup of sixteen waves -- so that
// between 2.6M and 10.5M rows per band ONE scatter level into 1024 big bins is enough: two passes over the keys instead of three (10M x 32: see
// launch_lsh_bucket_bands).  The bucket starts are 16-bit there (a bin holds fewer than 65 536 elements) to fit 4096 buckets beside the bin.
constexpr int kBigBinCap = 11264, kBigSubBits = 12, kBigSortThreads = 1024;
template <int kBinCap, int kSubBits, int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void lsh_bin_sort_kernel(const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ bin_start,
                                                           const uint64_t *__restrict__ slab_dig, const uint32_t *__restrict__ slab_row, int64_t n,
                                                           int32_t bands, int bin_bits, uint64_t *__restrict__ out_dig, uint32_t *__restrict__ out_row) {
    using StartT = typename std::conditional<(kBinCap > 4096), uint16_t, uint32_t>::type;
    static_assert(kBinCap < 65536, "16-bit bucket starts");
    __shared__ uint64_t dig[kBinCap];
    __shared__ uint32_t row[kBinCap];
    __shared__ uint32_t cnt[1 << kSubBits];
    __shared__ StartT start[1 << kSubBits];
    __shared__ uint32_t scan_tmp[kSortThreads / 64];
    const int nb = 1 << bin_bits, tid = threadIdx.x;
    constexpr int kSub = 1 << kSubBits, kPer = kSub / kSortThreads;
    for (int64_t item = blockIdx.x; item < (int64_t)bands * nb; item += gridDim.x) {
        const int64_t band = item >> bin_bits;
        const uint32_t count = min(cursor[item], (uint32_t)kBinCap);
        const int64_t out_base = band * n + bin_start[item];  // behind the band's bins before it (lsh_bin_offsets_kernel)
        for (int t = tid; t < kSub; t += kSortThreads) cnt[t] = 0;
        __syncthreads();
        const int64_t slab = item * kBinCap;
        const auto sub_of = [&](uint64_t d) { return (uint32_t)((bin_bits ? d << bin_bits : d) >> (64 - kSubBits)); };
        // bucket sizes first, then every element to its bucket's range in LDS -- one LDS copy of the bin, three workgroups per CU; a thread's
        // elements stay in its registers across the scan (until round 6 they came from the L2 a second time: -1.5 %)
        // (a thread's loads all go out before the first is waited for -- a thread behind the bin's end reads its last element again --
        // and the LDS atomics follow in a loop of their own: with the atomic next to its load the compiler waited for every load
        // in turn, kMine = kBinCap / kSortThreads = six (big form: eleven) memory latencies per bin and pass)
        constexpr int kMine = (kBinCap + kSortThreads - 1) / kSortThreads;
        const uint32_t last = count ? count - 1 : 0;
        uint64_t d[kMine];
        uint32_t rw[kMine];
        {
#pragma unroll
            for (int u = 0; u < kMine; ++u) {
                const uint32_t i = tid + u * kSortThreads;
                d[u] = slab_dig[slab + (i < count ? i : last)];
            }
#pragma unroll
            for (int u = 0; u < kMine; ++u) rw[u] = slab_row[slab + (tid + u * kSortThreads < count ? tid + u * kSortThreads : last)];  // (needed behind the scan: on their way meanwhile)
#pragma unroll
            for (int u = 0; u < kMine; ++u)
                if (tid + u * kSortThreads < count) atomicAdd(&cnt[sub_of(d[u])], 1u);
        }
        __syncthreads();
        // exclusive scan of the kSub bucket sizes (1024, or 4096 in the big form): a thread's kPer buckets (2 / 4), then the threads' sums
        uint32_t mine[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            mine[j] = cnt[tid * kPer + j];
            sum += mine[j];
        }
        const uint32_t incl = block_inclusive_scan(sum, scan_tmp, tid);
        uint32_t at = incl - sum;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            start[tid * kPer + j] = (StartT)at;
            cnt[tid * kPer + j] = 0;
            at += mine[j];
        }
        __syncthreads();
        {
#pragma unroll
            for (int u = 0; u < kMine; ++u)
                if (tid + u * kSortThreads < count) {
                    const uint32_t b = sub_of(d[u]);
                    const uint32_t p = start[b] + atomicAdd(&cnt[b], 1u);
                    dig[p] = d[u];
                    row[p] = rw[u];
                }
        }
        __syncthreads();
        // every element finds its place inside its bucket by counting the bucket's smaller (digest, row) pairs -- one or two
        // comparisons for uniform digests, the bucket's size for a cluster of equal ones (spread over the whole workgroup:
        // an element is a thread's, whatever its bucket) -- and goes straight to its position in the output
        for (uint32_t i = tid; i < count; i += kSortThreads) {
            const uint64_t d = dig[i];
            const uint32_t rw = row[i];
            const uint32_t b = sub_of(d), lo = start[b], hi = lo + cnt[b];
            uint32_t rank = 0;
            for (uint32_t j = lo; j < hi; ++j) rank += pair_less(dig[j], row[j], d, rw) ? 1u : 0u;
            out_dig[out_base + lo + rank] = d;
            out_row[out_base + lo + rank] = rw;
        }
        __syncthreads();
    }
}

// ---- the whole digest rides through the sort (n <= 2^(32 - band_bits)) --------------------------------
// The radix sort orders by the low sort_bits of a 64-bit key and moves the whole key and a 32-bit value every pass.
// The key's upper 64 - sort_bits bits and the value's bits above the row number are free luggage: the digest bits
// below the sorted prefix go there (the next 64 - sort_bits bits into the key, the last band_bits bits into the value
// under the row), so that after the sort the full digests are rebuilt from the sorted keys and values in one
// streaming pass -- instead of gathering them from the unsorted matrix by row (40 million random 8-byte reads: 0.8 ms
// of the 3.2 ms of a 32 x 1.25M sort).
__device__ __forceinline__ uint64_t low_bits(uint64_t x, int bits) { return bits >= 64 ? x : (bits <= 0 ? 0 : x & ((1ull << bits) - 1ull)); }

// (fused with the digests: the [n, bands] digest matrix is never written or read)
template <typename SigT>
__global__ __launch_bounds__(256) void band_keys_with_luggage_kernel(const SigT *__restrict__ sig, int32_t k, int32_t r, int64_t n, int32_t bands,
                                                                     int band_bits, int sort_bits, uint64_t *__restrict__ keys,
                                                                     uint32_t *__restrict__ vals) {
    const int64_t total = n * (int64_t)bands;
    const int d_bits = sort_bits - band_bits;   // digest bits inside the sorted prefix
    const int h_bits = 64 - sort_bits;          // digest bits that ride in the key's upper part
    const int shift = (bands & (bands - 1)) == 0 ? __builtin_ctz((unsigned)bands) : -1;  // a 64-bit division per element otherwise
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = shift >= 0 ? idx >> shift : idx / bands;
        const uint64_t dg = band_digest_of<SigT>(sig, row, (int)(idx - row * bands), k, r, n);
        const uint64_t prefix = prefix_key(dg, (uint32_t)(idx - row * bands), band_bits, sort_bits);
        const uint64_t hi = h_bits > 0 ? low_bits(dg >> band_bits, h_bits) : 0;  // bits [band_bits, band_bits + h_bits)
        keys[idx] = prefix | (h_bits > 0 ? hi << sort_bits : 0);
        vals[idx] = ((uint32_t)row << band_bits) | (uint32_t)low_bits(dg, band_bits);
        (void)d_bits;
    }
}

// keys and sorted_keys_only are the same buffer at the call site (read, then overwritten, element by element): no __restrict__
__global__ __launch_bounds__(256) void unpack_luggage_kernel(const uint64_t *keys, const uint32_t *__restrict__ vals,
                                                             int64_t total, int band_bits, int sort_bits,
                                                             uint64_t *sorted_keys_only, uint64_t *__restrict__ sorted_digests,
                                                             uint32_t *__restrict__ sorted_rows) {
    const int d_bits = sort_bits - band_bits, h_bits = 64 - sort_bits;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[p];
        const uint32_t val = vals[p];
        const uint64_t top = low_bits(key, d_bits);                                  // digest bits [64 - d_bits, 64)
        const uint64_t mid = h_bits > 0 ? key >> sort_bits : 0;                      // digest bits [band_bits, 64 - d_bits)
        uint64_t dg = low_bits(val, band_bits) | (mid << band_bits);
        if (d_bits > 0) dg |= top << (64 - d_bits);
        sorted_digests[p] = dg;
        sorted_rows[p] = val >> band_bits;
        sorted_keys_only[p] = low_bits(key, sort_bits);  // what the clean-up kernels compare: the sorted prefix alone
    }
}

// Runs of equal sort keys whose full digests are NOT all equal (different digests sharing the sorted
// prefix) have to be put in digest order.  Found in parallel: an element that has its predecessor's key but
// not its digest walks back to the head of its run and marks it; buckets of equal digests -- however large
// -- are never walked.
__global__ __launch_bounds__(256) void mark_mixed_runs_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ digests,
                                                              int64_t n, int64_t total, uint8_t *__restrict__ mixed) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        if (p % n == 0 || keys[p - 1] != keys[p] || digests[p - 1] == digests[p]) continue;
        const uint64_t key = keys[p];  // the band sits in the key: a run never crosses bands
        const int64_t band_start = p / n * n;
        int64_t head = p - 1;
        while (head > band_start && keys[head - 1] == key) --head;
        mixed[head] = 1;
    }
}

// The head of a marked run insertion-sorts it by digest -- stable, so rows stay ascending within equal digests.
__global__ __launch_bounds__(256) void order_mixed_runs_kernel(const uint64_t *__restrict__ keys, const uint8_t *__restrict__ mixed,
                                                               int64_t n, int64_t total, uint64_t *__restrict__ digests,
                                                               uint32_t *__restrict__ rows) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        if (!mixed[p]) continue;
        const uint64_t key = keys[p];
        const int64_t band_end = (p / n + 1) * n;
        int64_t end = p + 1;
        while (end < band_end && keys[end] == key) ++end;
        for (int64_t i = p + 1; i < end; ++i) {
            const uint64_t d = digests[i];
            const uint32_t r = rows[i];
            int64_t j = i;
            while (j > p && digests[j - 1] > d) {
                digests[j] = digests[j - 1];
                rows[j] = rows[j - 1];
                --j;
            }
            digests[j] = d;
            rows[j] = r;
        }
    }
}

// ---- candidate pairs from the sorted bands --------------------------------------------------------
// A bucket is a run of equal digests inside one band.  Element p of a run that starts at s pairs with
// the p - s elements in front of it, so the run of length L yields L(L-1)/2 pairs, each exactly once.
// Most elements are alone in their bucket: only an element that equals its predecessor looks for the
// start of its run (binary search in the sorted band, ~log2 n reads).

// ahead[p] = number of earlier elements of p's run (0 for a run's first element)
__global__ __launch_bounds__(256) void run_position_kernel(const uint64_t *__restrict__ digests, int64_t n, int64_t total,
                                                           uint32_t *__restrict__ ahead) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t band_start = p / n * n;
        uint32_t c = 0;
        if (p > band_start && digests[p] == digests[p - 1]) {
            const uint64_t d = digests[p];
            int64_t lo = band_start, hi = p - 1;  // first index in [band_start, p-1] holding d
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (digests[mid] < d) lo = mid + 1; else hi = mid;
            }
            c = (uint32_t)(p - lo);
        }
        ahead[p] = c;
    }
}

// raw[where[p] + q] = (min(row_p, row_q) << 32) | max(row_p, row_q) for the ahead[p] elements q in front of p
__global__ __launch_bounds__(256) void emit_pairs_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ ahead,
                                                         const uint64_t *__restrict__ where, int64_t total,
                                                         uint64_t *__restrict__ raw) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t c = ahead[p];
        if (c == 0) continue;
        const uint32_t me = rows[p];
        uint64_t *dst = raw + where[p];
        for (uint32_t q = 0; q < c; ++q) {
            const uint32_t other = rows[p - c + q];
            const uint32_t lo = me < other ? me : other, hi = me < other ? other : me;
            dst[q] = ((uint64_t)lo << 32) | hi;
        }
    }
}

__global__ __launch_bounds__(256) void unpack_pairs_kernel(const uint64_t *__restrict__ keys, int64_t count,
                                                           int64_t *__restrict__ pairs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        longlong2 v;
        v.x = (long long)(key >> 32);
        v.y = (long long)(key & 0xFFFFFFFFu);
        reinterpret_cast<longlong2 *>(pairs)[i] = v;
    }
}


// ---- bulk query against sorted bands ---------------------------------------------------------------
// What MinHashLSH.query does per probe (ref: datasketch/lsh.py:423-431: for every band, look the band key up
// in that band's dictionary and union the buckets), for M probes at once against an index of n rows held as
// sorted bands: the probe's band digest is located by binary search in the band's ascending digests; the
// matching run is its bucket.

// per (probe q, band j): first[idx] = position of the first equal digest in the band, count[idx] = run length
__global__ __launch_bounds__(256) void query_ranges_kernel(const uint64_t *__restrict__ q_digests, int64_t m, int32_t bands,
                                                           const uint64_t *__restrict__ sorted_digests, int64_t n,
                                                           uint32_t *__restrict__ first, uint32_t *__restrict__ count) {
    const int64_t total = m * (int64_t)bands;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int band = (int)(idx % bands);
        const uint64_t d = q_digests[idx];
        const uint64_t *col = sorted_digests + (int64_t)band * n;
        int64_t lo = 0, hi = n;  // lower bound
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (col[mid] < d) lo = mid + 1; else hi = mid;
        }
        int64_t end = lo;
        if (lo < n && col[lo] == d) {  // upper bound by galloping: buckets are short
            int64_t step = 1;
            end = lo + 1;
            while (end < n && col[end] == d) {
                end = std::min<int64_t>(n, end + step);
                step <<= 1;
            }
            int64_t a = std::max<int64_t>(lo, end - step / 2 - 1), b = end;  // last equal is in [a, b)
            while (a < b) {
                const int64_t mid = (a + b) >> 1;
                if (col[mid] <= d) a = mid + 1; else b = mid;
            }
            end = a;
        }
        first[idx] = (uint32_t)lo;
        count[idx] = (uint32_t)(end - lo);
    }
}

// raw[where[idx] + i] = (q << 32) | row for the rows of the probe's bucket in band j.  With VERIFY the r words of
// the band are compared (probe signature against index signature): a 64-bit digest collision between different
// band keys then yields no candidate -- exactly the reference's dictionary semantics -- and the slot gets ~0.
template <typename SigT, bool VERIFY>
__global__ __launch_bounds__(256) void query_emit_kernel(const uint32_t *__restrict__ first, const uint32_t *__restrict__ count,
                                                         const uint64_t *__restrict__ where, int64_t m, int32_t bands, int64_t n,
                                                         const uint32_t *__restrict__ sorted_rows,
                                                         const SigT *__restrict__ q_sig, const SigT *__restrict__ idx_sig,
                                                         int32_t k, int32_t r, uint64_t *__restrict__ raw) {
    const int64_t total = m * (int64_t)bands;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t c = count[idx];
        if (c == 0) continue;
        const int64_t q = idx / bands;
        const int band = (int)(idx - q * bands);
        const uint32_t *rows = sorted_rows + (int64_t)band * n + first[idx];
        uint64_t *dst = raw + where[idx];
        for (uint32_t i = 0; i < c; ++i) {
            const uint32_t row = rows[i];
            bool same = true;
            if (VERIFY) {
                const SigT *x = q_sig + q * k + (int64_t)band * r, *y = idx_sig + (int64_t)row * k + (int64_t)band * r;
                for (int w = 0; w < r; ++w) same &= x[w] == y[w];
            }
            dst[i] = same ? (((uint64_t)q << 32) | row) : ~0ull;
        }
    }
}

// ---- device-wide exclusive scan, hand-written (round 5: rocPRIM's exclusive_scan and unique are gone from this file) ------
// Three launches over tiles of 256 threads x 16 items: (1) every tile's sum, (2) one workgroup turns the tile sums into tile
// offsets (and the grand total), (3) every tile scans again from its offset and hands (index, exclusive prefix, value) to the
// output functor.  The input is a functor too, so that "unique" is the same three launches: value = 1 where a sorted key
// differs from its predecessor, output = the key written at its prefix.  The input is read twice (8 B + 8 B per element for
// 40M counts -> where: 0.5 GB, ~0.15 ms); a decoupled look-back would read it once and is not worth its spin loops here.
constexpr int kScanItems = 16, kScanTile = 256 * kScanItems;

struct CountsIn {  // value = counts[i]
    const uint32_t *counts;
    __device__ __forceinline__ uint32_t get(int64_t i) const { return counts[i]; }
};
struct WhereOut {  // where[i] = exclusive prefix
    uint64_t *where;
    __device__ __forceinline__ void put(int64_t i, uint64_t prefix, uint32_t) const { where[i] = prefix; }
};
struct HeadsIn {  // value = 1 at the first element of a run of equal sorted keys
    const uint64_t *keys;
    __device__ __forceinline__ uint32_t get(int64_t i) const { return i == 0 || keys[i] != keys[i - 1] ? 1u : 0u; }
};
struct CompactOut {  // the run heads, packed
    const uint64_t *keys;
    uint64_t *out;
    __device__ __forceinline__ void put(int64_t i, uint64_t prefix, uint32_t head) const {
        if (head) out[prefix] = keys[i];
    }
};

__device__ __forceinline__ uint64_t block_inclusive_scan64(uint64_t v, uint64_t *tmp4, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, o), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), o);
        if (lane >= o) v += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) tmp4[wave] = v;
    __syncthreads();
    uint64_t add = 0;
    for (int w = 0; w < wave; ++w) add += tmp4[w];
    __syncthreads();
    return v + add;
}

template <typename In>
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(In in, int64_t n, uint64_t *__restrict__ tile_sums) {
    __shared__ uint64_t tmp[4];
    const int64_t first = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint64_t sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j)
        if (first + j < n) sum += in.get(first + j);
    const uint64_t incl = block_inclusive_scan64(sum, tmp, threadIdx.x);
    if (threadIdx.x == 255) tile_sums[blockIdx.x] = incl;
}

// tile sums -> exclusive tile offsets in place; total[0] = the grand total.  One workgroup: a 10M-row index has 80 000 tiles.
__global__ __launch_bounds__(1024) void scan_tile_offsets_kernel(uint64_t *__restrict__ tiles, int64_t n_tiles, uint64_t *__restrict__ total) {
    __shared__ uint64_t tmp[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_tiles; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint64_t v = i < n_tiles ? tiles[i] : 0;
        // inclusive scan over the 1024 threads: 16 waves
        uint64_t x = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)x, o), hi = (uint32_t)__shfl_up((int)(uint32_t)(x >> 32), o);
            if (lane >= o) x += ((uint64_t)hi << 32) | lo;
        }
        if (lane == 63) tmp[wave] = x;
        __syncthreads();
        uint64_t add = carry;
        for (int w = 0; w < wave; ++w) add += tmp[w];
        if (i < n_tiles) tiles[i] = add + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = add + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry;
}

template <typename In, typename Out>
__global__ __launch_bounds__(256) void scan_apply_kernel(In in, int64_t n, const uint64_t *__restrict__ tile_offsets, Out out) {
    __shared__ uint64_t tmp[4];
    const int64_t first = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint64_t sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        v[j] = first + j < n ? in.get(first + j) : 0u;
        sum += v[j];
    }
    uint64_t at = tile_offsets[blockIdx.x] + block_inclusive_scan64(sum, tmp, threadIdx.x) - sum;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        if (first + j < n) out.put(first + j, at, v[j]);
        at += v[j];
    }
}

// scratch words the scan needs for n elements: the tile sums and the total
inline size_t scan_tmp_bytes(int64_t n) { return ((sizeof(uint64_t) * (size_t)((n + kScanTile - 1) / kScanTile + 2)) + 255) & ~(size_t)255; }

// enqueues the three launches; the grand total lands in d_tmp[n_tiles] (device) -- the caller reads it back when it needs it
template <typename In, typename Out>
int device_exclusive_scan(mhx_ctx *ctx, In in, Out out, int64_t n, void *d_tmp, uint64_t **d_total) {
    const int64_t n_tiles = (n + kScanTile - 1) / kScanTile;
    uint64_t *tiles = static_cast<uint64_t *>(d_tmp);
    *d_total = tiles + n_tiles;
    if (n_tiles >= (int64_t)1 << 31) return fail(MHX_ERR_UNSUPPORTED, "scan of more than 2^43 elements");
    hipLaunchKernelGGL(scan_tile_sums_kernel<In>, dim3((unsigned)n_tiles), dim3(256), 0, ctx->stream, in, n, tiles);
    hipLaunchKernelGGL(scan_tile_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, tiles, n_tiles, *d_total);
    hipLaunchKernelGGL((scan_apply_kernel<In, Out>), dim3((unsigned)n_tiles), dim3(256), 0, ctx->stream, in, n, tiles, out);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

unsigned grid_for(const mhx_ctx *ctx, int64_t items) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((items + 255) / 256, (int64_t)ctx->num_cus * 16));
}

}  // namespace

int launch_lsh_candidate_pairs(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n,
                               int32_t bands, int64_t *d_pairs, int64_t capacity, int64_t *n_pairs, int64_t *n_raw) {
    *n_pairs = 0;
    if (n_raw) *n_raw = 0;
    const int64_t total = n * (int64_t)bands;
    if (total == 0) return MHX_OK;
    // scratch[4], first part: ahead u32[total] | where u64[total] | tail u64[2] | scan temporary
    const size_t ahead_bytes = ((sizeof(uint32_t) * (size_t)total) + 255) & ~(size_t)255;
    const size_t where_bytes = ((sizeof(uint64_t) * (size_t)total) + 255) & ~(size_t)255;
    const size_t scan_tmp = scan_tmp_bytes(total);
    if (int rc = ctx->ensure_scratch(4, ahead_bytes + where_bytes + 256 + scan_tmp)) return rc;
    uint32_t *d_ahead = (uint32_t *)ctx->scratch[4];
    uint64_t *d_where = (uint64_t *)((char *)ctx->scratch[4] + ahead_bytes);
    void *d_scan_tmp = (char *)ctx->scratch[4] + ahead_bytes + where_bytes + 256;
    hipLaunchKernelGGL(run_position_kernel, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, d_sorted_digests, n, total,
                       d_ahead);
    MHX_HIP_CHECK(hipGetLastError());
    uint64_t *d_raw_total = nullptr;
    if (int rc = device_exclusive_scan(ctx, CountsIn{d_ahead}, WhereOut{d_where}, total, d_scan_tmp, &d_raw_total)) return rc;
    uint64_t raw_total = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&raw_total, d_raw_total, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int64_t raw = (int64_t)raw_total;  // pairs before deduplication across bands
    if (n_raw) *n_raw = raw;
    if (raw == 0) return MHX_OK;
    if ((size_t)raw * 16 > (size_t)ctx->hbm_bytes / 2)
        return fail(MHX_ERR_OOM, "%lld candidate pairs before deduplication (large buckets of equal band keys) do not fit in device memory",
                    (long long)raw);

    // scratch[3]: raw u64[raw] | sorted u64[raw] | count u64 | sort / select temporary
    const size_t raw_bytes = ((sizeof(uint64_t) * (size_t)raw) + 255) & ~(size_t)255;
    int end_bit = 33;  // the high word holds a row number < n
    while (end_bit < 64 && ((int64_t)1 << (end_bit - 32)) < n) ++end_bit;
    size_t sort_tmp = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)raw, 0, end_bit,
                                            ctx->stream);  // (the one library primitive left on this path: a radix sort of the raw pairs)
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim size query failed: %s", hipGetErrorString(e));
    const size_t tmp_bytes = std::max(sort_tmp, scan_tmp_bytes(raw));
    if (int rc = ctx->ensure_scratch(3, 2 * raw_bytes + 256 + tmp_bytes)) return rc;
    uint64_t *d_raw = (uint64_t *)ctx->scratch[3];
    uint64_t *d_sorted = (uint64_t *)((char *)ctx->scratch[3] + raw_bytes);
    void *d_tmp = (char *)ctx->scratch[3] + 2 * raw_bytes + 256;
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, d_sorted_rows, d_ahead, d_where,
                       total, d_raw);
    MHX_HIP_CHECK(hipGetLastError());
    e = rocprim::radix_sort_keys(d_tmp, sort_tmp, (const uint64_t *)d_raw, d_sorted, (size_t)raw, 0, end_bit, ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_keys failed: %s", hipGetErrorString(e));
    // unique: the heads of the runs of equal sorted pairs, packed (the scan above with other functors)
    uint64_t *d_count = nullptr;
    if (int rc = device_exclusive_scan(ctx, HeadsIn{d_sorted}, CompactOut{d_sorted, d_raw}, raw, d_tmp, &d_count)) return rc;
    uint64_t unique_count = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&unique_count, d_count, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_pairs = (int64_t)unique_count;
    if ((int64_t)unique_count > capacity) return MHX_OK;  // caller sees n_pairs > capacity and calls again
    hipLaunchKernelGGL(unpack_pairs_kernel, dim3(grid_for(ctx, (int64_t)unique_count)), dim3(256), 0, ctx->stream, d_raw,
                       (int64_t)unique_count, d_pairs);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

// the two-pass bucketing; *done = false when a bin overflowed (the caller falls back to the radix sort)
// [n, bands] digests -> [bands, n] (bands a power of two <= 64): a workgroup takes 2048 consecutive (row, band) pairs -- whole rows,
// read with unit stride --, parks them in LDS as [band][row of the tile] (one word of padding per band: the lanes of a wave differ
// in the band first) and writes every band's run of the tile with consecutive lanes.  In front of the bucketing of a ROW-major
// digest matrix: its scatter pass reads a band with a stride of `bands` words, every 128-byte line fetched by the four XCDs whose
// bands share it (1.30 GB of reads for 320 MB), and with all of a thread's loads in flight at once that costs more than this pass
// (0.64 GB of traffic) and the unit-stride bucketing behind it together.
__global__ __launch_bounds__(256) void digests_to_band_major_kernel(const uint64_t *__restrict__ in, int64_t n, int shift, uint64_t *__restrict__ out) {
    constexpr int kTile = 2048;
    __shared__ uint64_t tile[kTile + 64];
    const int bands = 1 << shift, tr = kTile >> shift;
    const int64_t total = n << shift;
    for (int64_t base = (int64_t)blockIdx.x * kTile; base < total; base += (int64_t)gridDim.x * kTile) {
        uint64_t v[kTile / 256];
#pragma unroll
        for (int it = 0; it < kTile / 256; ++it) {
            const int64_t idx = base + it * 256 + threadIdx.x;
            v[it] = in[idx < total ? idx : total - 1];
        }
#pragma unroll
        for (int it = 0; it < kTile / 256; ++it) {
            const int local = it * 256 + threadIdx.x;
            tile[(local & (bands - 1)) * (tr + 1) + (local >> shift)] = v[it];
        }
        __syncthreads();
        const int64_t row0 = base >> shift;
        for (int e = threadIdx.x; e < kTile; e += 256) {
            const int band = e / tr, row = e - band * tr;
            if (row0 + row < n) out[(int64_t)band * n + row0 + row] = tile[band * (tr + 1) + row];
        }
        __syncthreads();
    }
}

// one scatter pass (see lsh_bin_scatter_kernel); returns false when the launch is refused
static size_t scatter_team_bytes(int lo_bits, bool pairs, int rows, int team = 256) {
    const size_t nb = (size_t)1 << lo_bits, chunk = (size_t)team * (size_t)rows;
    return 8 * chunk + 8 * ((3 * nb * 4 + chunk * (pairs ? 4 : 2) + 64 + 7) / 8);
}

template <typename SigT, int ROWS>
static bool launch_scatter_rows(mhx_ctx *ctx, const SigT *d_sig, int32_t k, int32_t r, int64_t n, int32_t units, int hi_bits, int lo_bits, int band_share,
                                uint32_t cap, uint32_t *d_cursor, uint64_t *d_slab_dig, uint32_t *d_slab_row, uint32_t *d_overflow, const uint64_t *d_src_dig,
                                const uint32_t *d_src_row, const uint32_t *d_src_cursor, uint32_t src_cap) {
    constexpr bool kPairs = std::is_same<SigT, SlabPairs>::value;
    const int64_t span = kPairs ? (int64_t)src_cap : n;
    if constexpr (ROWS == 16 && (kPairs || std::is_same<SigT, Digest64BM>::value)) {
        // Round 6: a unit-stride source is taken by ONE team of 1024 threads x 8 rows per workgroup (92 KB of LDS: one workgroup, sixteen waves
        // per CU) instead of three teams of 256 x 16 -- 8192 rows per (team, bin) histogram: half the returning cursor atomics (one per
        // (team, bin): 5M instead of 10M for 40M keys) and runs of eight elements instead of four for the L2 to merge into lines.  Same box,
        // interleaved (tools/ab_option.py lsh.team): 40M keys 0.472 -> 0.441 ms, 320M keys 5.08 -> 4.87 ms; 512 x 16: 0.457 / 4.99, 768 x 16:
        // 0.470 / 5.04, 1024 x 12: 0.445 / 4.89, 1024 x 14: 0.443 / 5.19, 512 x 24 rows: 0.54 / 7.4 (profiles/r06_ab_scatter_team.txt).
        // Option lsh.team 256: the teams of 256.
        constexpr int TT = 1024, RR = 8;
        const size_t lds2 = scatter_team_bytes(lo_bits, kPairs, RR, TT);
        const int64_t items2 = (span + TT * RR - 1) / (TT * RR) * units;
        // (fewer than eight items per CU -- 200k rows x 32 bands: 0.101 against 0.095 ms -- leave the single workgroup per CU short of work: teams of 256)
        if (band_share == 1 && ctx->opt_lsh_team != 256 && lds2 <= (size_t)ctx->lds_per_block && (items2 >= 8 * (int64_t)ctx->num_cus || ctx->opt_lsh_team == 1024)) {
            const int64_t per_cu2 = std::max<int64_t>(1, (int64_t)((size_t)ctx->lds_per_block / (lds2 + 64)));
            const unsigned grid2 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(items2, (int64_t)ctx->num_cus * per_cu2 * 2));
            hipLaunchKernelGGL((lsh_bin_scatter_kernel<SigT, RR, TT>), dim3(grid2), dim3(TT), lds2, ctx->stream, d_sig, k, r, n, units, hi_bits, lo_bits, 1, cap,
                               d_cursor, d_slab_dig, d_slab_row, d_overflow, d_src_dig, d_src_row, d_src_cursor, src_cap);
            return hipGetLastError() == hipSuccess;
        }
    }
    const int64_t items = (span + 256 * ROWS - 1) / (256 * ROWS) * (units / band_share);
    const size_t lds = scatter_team_bytes(lo_bits, kPairs, ROWS) * band_share;
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(32 / (4 * band_share), (int64_t)((size_t)ctx->lds_per_block / (lds + 64))));
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(items, (int64_t)ctx->num_cus * per_cu * 2));
    hipLaunchKernelGGL((lsh_bin_scatter_kernel<SigT, ROWS>), dim3(grid), dim3(256 * band_share), lds, ctx->stream, d_sig, k, r, n, units, hi_bits, lo_bits, band_share, cap,
                       d_cursor, d_slab_dig, d_slab_row, d_overflow, d_src_dig, d_src_row, d_src_cursor, src_cap);
    return hipGetLastError() == hipSuccess;
}

// rows per thread: 16 for the sources a team reads with unit stride (band-major digests, the big bins of level 0) -- twice the run
// per (team, bin), half the atomics per key, 46 KB of LDS per one-team workgroup: 0.714 -> 0.664 ms for 40M keys, same box -- and 8
// where four teams share the lines of a signature matrix (four times 46 KB would not fit); option lsh.chunk = 8 forces eight
static int scatter_rows(const mhx_ctx *ctx, bool unit_stride) { return unit_stride && ctx->opt_lsh_chunk != 8 ? 16 : kScatterRowsDefault; }

template <typename SigT>
static bool launch_scatter(mhx_ctx *ctx, const SigT *d_sig, int32_t k, int32_t r, int64_t n, int32_t units, int hi_bits, int lo_bits, int band_share,
                           uint32_t cap, uint32_t *d_cursor, uint64_t *d_slab_dig, uint32_t *d_slab_row, uint32_t *d_overflow, const uint64_t *d_src_dig,
                           const uint32_t *d_src_row, const uint32_t *d_src_cursor, uint32_t src_cap) {
    constexpr bool kUnit = std::is_same<SigT, SlabPairs>::value || std::is_same<SigT, Digest64BM>::value;
    if (scatter_rows(ctx, kUnit) == 16)
        return launch_scatter_rows<SigT, 16>(ctx, d_sig, k, r, n, units, hi_bits, lo_bits, band_share, cap, d_cursor, d_slab_dig, d_slab_row, d_overflow, d_src_dig,
                                             d_src_row, d_src_cursor, src_cap);
    return launch_scatter_rows<SigT, 8>(ctx, d_sig, k, r, n, units, hi_bits, lo_bits, band_share, cap, d_cursor, d_slab_dig, d_slab_row, d_overflow, d_src_dig, d_src_row,
                                        d_src_cursor, src_cap);
}

// the two-pass (or, beyond 2^10 bins per band, three-pass) bucketing; *done = false when a bin overflowed or a resource could not
// be had (the caller falls back to the radix sort, which handles every size)
static int launch_lsh_bucket_bands(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands, int32_t r,
                                   uint64_t *d_sorted_digests, uint32_t *d_sorted_rows, bool *done) {
    *done = false;
    int bin_bits = 0;
    while (bin_bits < kMaxBinBits && (n >> bin_bits) > 2500) ++bin_bits;  // about 1250 .. 2500 elements per bin (kBinCap: 3072)
    if ((n >> bin_bits) > 2500) return MHX_OK;                            // more than 41 million rows: the radix sort
    // Round 6: between 2.56M and 10.2M rows ONE scatter level into 2^10 big bins of up to kBigBinCap elements, finished by the bin pass's
    // <kBigBinCap, 12, 1024> form (a CU's whole LDS): two passes over the keys instead of three.  Option lsh.bigbins: 1 = never, 2 = from 4 bins on (tests).
    const bool big = ctx->opt_lsh_bigbins != 1 && ctx->opt_lsh_levels != 2 &&
                     ((bin_bits > kOneLevelBits && (n >> kOneLevelBits) <= (kBigBinCap * 10) / 11) || (ctx->opt_lsh_bigbins == 2 && bin_bits >= 2));
    if (big) bin_bits = ctx->opt_lsh_bigbins == 2 && bin_bits <= kOneLevelBits ? bin_bits - 2 : kOneLevelBits;
    const int32_t bin_cap = big ? kBigBinCap : kBinCap;
    // beyond 2^10 bins: two scatter levels of about half the bits each (see the kernel)
    const int hi_bits = bin_bits > kOneLevelBits || (ctx->opt_lsh_levels == 2 && bin_bits >= 2) ? bin_bits / 2 : 0, lo_bits = bin_bits - hi_bits;  // (option lsh.levels = 2: two levels at any size, for the tests)
    const int64_t nb = (int64_t)1 << bin_bits, bins = nb * bands;
    const int64_t big_bins = hi_bits ? ((int64_t)bands << hi_bits) : 0;
    const uint32_t cap0 = hi_bits ? (uint32_t)std::min<int64_t>(0xFFFFFFFFll, (n >> hi_bits) + (n >> hi_bits) / 32 + 4096) : 0;  // 3 % + 4096 over the mean (sigma = sqrt(mean))
    const size_t cur_bytes = ((sizeof(uint32_t) * (size_t)(2 * bins + big_bins + 2)) + 255) & ~(size_t)255;  // cursor[bins] | overflow | cursor0[big_bins] | overflow0 | bin_start[bins]: 2 * bins + big_bins + 2 words (both overflow words counted)
    const size_t dig_bytes = sizeof(uint64_t) * (size_t)bins * bin_cap, row_bytes = sizeof(uint32_t) * (size_t)bins * bin_cap;
    const size_t dig0_bytes = ((sizeof(uint64_t) * (size_t)big_bins * cap0) + 255) & ~(size_t)255, row0_bytes = ((sizeof(uint32_t) * (size_t)big_bins * cap0) + 255) & ~(size_t)255;
    if (cur_bytes + dig_bytes + row_bytes + dig0_bytes + row0_bytes > (size_t)ctx->hbm_bytes / 4) return MHX_OK;
    // bands whose r values of a row share a 128-byte line go to one workgroup (at most four) -- as far as the teams'
    // staging areas fit the LDS of a workgroup; not even one team fitting, a slab that cannot be had, a launch that is
    // refused: the radix sort handles every size
    const int piece = sig_dtype == kSigDigestsBM ? 128 : r * (sig_dtype == MHX_U32 ? 4 : 8);  // (digests: r = 1, 8 bytes; band-major digests: a team's rows are contiguous, bands share nothing)
    int band_share = piece < 128 && 128 % piece == 0 ? std::min(4, 128 / piece) : 1;
    while (bands % band_share) band_share >>= 1;
    const int first_bits = hi_bits ? hi_bits : lo_bits;  // bits of the pass that reads the signatures
    const size_t team_bytes = scatter_team_bytes(first_bits, false, scatter_rows(ctx, sig_dtype == kSigDigestsBM));
    const size_t lds_limit = (size_t)ctx->lds_per_block;
    while (band_share > 1 && team_bytes * band_share > lds_limit) band_share >>= 1;
    if (team_bytes * band_share > lds_limit || scatter_team_bytes(lo_bits, true, scatter_rows(ctx, true)) > lds_limit) return MHX_OK;
    if (ctx->ensure_scratch(3, cur_bytes + dig_bytes + row_bytes + dig0_bytes + row0_bytes + 256) != MHX_OK) return MHX_OK;
    char *base = (char *)ctx->scratch[3];
    uint32_t *d_cursor = (uint32_t *)base;
    uint32_t *d_overflow = d_cursor + bins;
    uint32_t *d_cursor0 = d_overflow + 1;
    uint32_t *d_overflow0 = d_cursor0 + big_bins;
    uint32_t *d_bin_start = d_overflow0 + 1;
    uint64_t *d_slab_dig = (uint64_t *)(base + cur_bytes);
    uint32_t *d_slab_row = (uint32_t *)(base + cur_bytes + dig_bytes);
    uint64_t *d_slab0_dig = (uint64_t *)(base + cur_bytes + dig_bytes + row_bytes);
    uint32_t *d_slab0_row = (uint32_t *)(base + cur_bytes + dig_bytes + row_bytes + dig0_bytes);
    MHX_HIP_CHECK(hipMemsetAsync(d_cursor, 0, sizeof(uint32_t) * (size_t)(bins + big_bins + 2), ctx->stream));
    // the pass that reads the signatures: into the final slabs (one level) or into the big bins (level 0 of two)
    uint32_t *cur_a = hi_bits ? d_cursor0 : d_cursor, *ovf_a = hi_bits ? d_overflow0 : d_overflow;
    uint64_t *dig_a = hi_bits ? d_slab0_dig : d_slab_dig;
    uint32_t *row_a = hi_bits ? d_slab0_row : d_slab_row;
    const uint32_t cap_a = hi_bits ? cap0 : (uint32_t)bin_cap;
    bool ok;
#define MHX_SCATTER_A(T) ok = launch_scatter<T>(ctx, (const T *)d_sig, k, r, n, bands, 0, first_bits, band_share, cap_a, cur_a, dig_a, row_a, ovf_a, nullptr, nullptr, nullptr, 0)
    if (sig_dtype == kSigDigestsBM) MHX_SCATTER_A(Digest64BM);
    else if (sig_dtype == kSigDigests) MHX_SCATTER_A(Digest64);
    else if (sig_dtype == MHX_U32) MHX_SCATTER_A(uint32_t);
    else MHX_SCATTER_A(uint64_t);
#undef MHX_SCATTER_A
    if (!ok) return MHX_OK;  // (a launch the device refuses: nothing has run, the radix sort takes over)
    if (hi_bits) {  // level 1: every big bin into its 2^lo final bins
        if (!launch_scatter<SlabPairs>(ctx, (const SlabPairs *)nullptr, k, r, n, (int32_t)big_bins, hi_bits, lo_bits, 1, (uint32_t)kBinCap, d_cursor, d_slab_dig, d_slab_row,
                                       d_overflow, d_slab0_dig, d_slab0_row, d_cursor0, cap0))
            return MHX_OK;
    }
    uint32_t overflow[2] = {0, 0};
    MHX_HIP_CHECK(hipMemcpyAsync(&overflow[0], d_overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (hi_bits) MHX_HIP_CHECK(hipMemcpyAsync(&overflow[1], d_overflow0, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (overflow[0] || overflow[1]) return MHX_OK;
    hipLaunchKernelGGL(lsh_bin_offsets_kernel, dim3((unsigned)bands), dim3(256), 0, ctx->stream, d_cursor, bin_bits, (uint32_t)bin_cap, d_bin_start);
    if (big)
        hipLaunchKernelGGL((lsh_bin_sort_kernel<kBigBinCap, kBigSubBits, kBigSortThreads>), dim3((unsigned)std::min<int64_t>(bins, (int64_t)ctx->num_cus * 32)), dim3(kBigSortThreads), 0,
                           ctx->stream, d_cursor, d_bin_start, d_slab_dig, d_slab_row, n, bands, bin_bits, d_sorted_digests, d_sorted_rows);
    else
        hipLaunchKernelGGL((lsh_bin_sort_kernel<kBinCap, kSubBits, kSortThreads>), dim3((unsigned)std::min<int64_t>(bins, (int64_t)ctx->num_cus * 96)), dim3(kSortThreads), 0, ctx->stream,
                           d_cursor, d_bin_start, d_slab_dig, d_slab_row, n, bands, bin_bits, d_sorted_digests, d_sorted_rows);
    MHX_HIP_CHECK(hipGetLastError());
    *done = true;
    return MHX_OK;
}

int launch_lsh_sort_bands(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands, int32_t r,
                          uint64_t *d_sorted_digests, uint32_t *d_sorted_rows) {
    if (n >= ((int64_t)1 << 32)) return fail(MHX_ERR_UNSUPPORTED, "more than 2^32-1 signatures per call");
    if (ctx->opt_lsh_sort != 1 && ctx->opt_lsh_sort_bits == 0 && ctx->opt_lsh_gather == 0 && n > 0) {  // the options of the radix path select it
        bool done = false;
        const void *d_in = d_sig;
        int in_dtype = sig_dtype;
        // The band-major copy of the digests (8 bytes per key in scratch[4], kept until mhx_ctx_release_scratch) is optional: it counts against
        // the same quarter of the device's memory as the bucketing's slabs (~30 bytes per key, launch_lsh_bucket_bands) -- 10M x 32: 2.5 of 12 GB --
        // and when it does not fit, or cannot be had, the bucketing reads the caller's matrix as it is and the call's error string stays empty.
        const size_t pre_bytes = sizeof(uint64_t) * (size_t)n * (size_t)bands;
        const bool pre_fits = pre_bytes + 30 * (size_t)n * (size_t)bands <= (size_t)ctx->hbm_bytes / 4;
        const auto pre_buffer = [&]() {
            if (!pre_fits) return false;
            if (ctx->ensure_scratch(4, pre_bytes) == MHX_OK) return true;
            forgive();
            return false;
        };
        if (sig_dtype == kSigDigests && bands >= 2 && bands <= 64 && (bands & (bands - 1)) == 0 && pre_buffer()) {
            // a row-major digest matrix is turned band-major first (see digests_to_band_major_kernel)
            const int64_t tiles = (n * bands + 2047) / 2048;
            hipLaunchKernelGGL(digests_to_band_major_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)ctx->num_cus * 16))), dim3(256), 0, ctx->stream,
                               (const uint64_t *)d_sig, n, __builtin_ctz((unsigned)bands), (uint64_t *)ctx->scratch[4]);
            if (hipGetLastError() == hipSuccess) d_in = ctx->scratch[4], in_dtype = kSigDigestsBM;
        } else if ((sig_dtype == MHX_U32 || sig_dtype == MHX_U64) && ctx->opt_lsh_prehash != 1 && bands >= 2 && bands <= 64 && (bands & (bands - 1)) == 0 &&
                   pre_buffer()) {
            // a signature matrix: its band digests first, band-major (band_digest_bm_kernel: one read of the matrix at the stream's rate), then the
            // bucketing of the digests -- hashing r values inside the scatter pass's load loop cost more than the extra 8 bytes per key written and
            // read again (1.25M x 256 uint32, 32 x 8: 0.90 ms against 0.31 + 0.48).  Option lsh.prehash 1: hash inside the scatter pass.
            if (launch_band_digests(ctx, d_sig, sig_dtype, n, k, bands, r, (uint64_t *)ctx->scratch[4], MHX_BAND_MAJOR) == MHX_OK) d_in = ctx->scratch[4], in_dtype = kSigDigestsBM;
        }
        if (int rc = launch_lsh_bucket_bands(ctx, d_in, in_dtype, n, k, bands, r, d_sorted_digests, d_sorted_rows, &done)) return rc;
        if (done) return MHX_OK;
    }
    // scratch[3]: digests[n, bands] | keys u64[total] | sorted keys u64[total] | rows u32[total] | marks u8[total] |
    // rocPRIM temporary
    const int64_t total = n * (int64_t)bands;
    const size_t dig_bytes = ((sizeof(uint64_t) * (size_t)total) + 255) & ~(size_t)255;
    const size_t row_bytes = ((sizeof(uint32_t) * (size_t)total) + 255) & ~(size_t)255;
    int band_bits = 1;
    while (((int64_t)1 << band_bits) < bands) ++band_bits;
    if (band_bits > 16) return fail(MHX_ERR_UNSUPPORTED, "more than 65536 bands");
    // bits handed to the radix sort: band + a digest prefix of log2(n) + 5 bits, rounded up to whole 8-bit passes -- about
    // n/64 elements per band then share a prefix with another digest and are left to the clean-up kernels, which is
    // cheaper than a fifth pass over all of them (32 bands x 1.25M rows: 32 bits 2.32 ms, 40 bits 2.64 ms, 24 bits 3.36 ms)
    int log_n = 1;
    while (((int64_t)1 << log_n) < n) ++log_n;
    const int auto_bits = std::min(64, (band_bits + std::max(12, log_n + 5) + 7) / 8 * 8);
    const int sort_bits = (int)std::min<int64_t>(64, std::max<int64_t>(band_bits, ctx->opt_lsh_sort_bits > 0 ? ctx->opt_lsh_sort_bits : auto_bits));
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                             (const uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)total, 0, sort_bits,
                                             ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed: %s", hipGetErrorString(e));
    const size_t mark_bytes = ((size_t)total + 255) & ~(size_t)255;
    if (int rc = ctx->ensure_scratch(3, 3 * dig_bytes + row_bytes + mark_bytes + tmp_bytes + 512)) return rc;
    uint64_t *d_dig = (uint64_t *)ctx->scratch[3];
    uint64_t *d_keys = (uint64_t *)((char *)ctx->scratch[3] + dig_bytes);
    uint64_t *d_keys_sorted = (uint64_t *)((char *)ctx->scratch[3] + 2 * dig_bytes);
    uint32_t *d_rows = (uint32_t *)((char *)ctx->scratch[3] + 3 * dig_bytes);
    uint8_t *d_mixed = (uint8_t *)((char *)ctx->scratch[3] + 3 * dig_bytes + row_bytes);
    void *d_tmp = (char *)ctx->scratch[3] + 3 * dig_bytes + row_bytes + mark_bytes;
    const dim3 grid(grid_for(ctx, total));
    const bool luggage = n <= ((int64_t)1 << (32 - band_bits)) && ctx->opt_lsh_gather != 1;  // the row and band_bits digest bits fit the 32-bit value
    if (luggage) {
        // the digest region receives the sorted values, d_sorted_rows the final rows
        if (sig_dtype == kSigDigestsBM)
            hipLaunchKernelGGL(band_keys_with_luggage_kernel<Digest64BM>, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, (const Digest64BM *)d_sig, k, r, n,
                               bands, band_bits, sort_bits, d_keys, d_rows);
        else if (sig_dtype == kSigDigests)
            hipLaunchKernelGGL(band_keys_with_luggage_kernel<Digest64>, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, (const Digest64 *)d_sig, k, r, n,
                               bands, band_bits, sort_bits, d_keys, d_rows);
        else if (sig_dtype == MHX_U32)
            hipLaunchKernelGGL(band_keys_with_luggage_kernel<uint32_t>, dim3(grid_for(ctx, total) ), dim3(256), 0, ctx->stream, (const uint32_t *)d_sig, k, r, n,
                               bands, band_bits, sort_bits, d_keys, d_rows);
        else
            hipLaunchKernelGGL(band_keys_with_luggage_kernel<uint64_t>, dim3(grid_for(ctx, total)), dim3(256), 0, ctx->stream, (const uint64_t *)d_sig, k, r, n,
                               bands, band_bits, sort_bits, d_keys, d_rows);
        MHX_HIP_CHECK(hipGetLastError());
        uint32_t *d_vals_sorted = reinterpret_cast<uint32_t *>(d_dig);
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, (const uint64_t *)d_keys, d_keys_sorted, (const uint32_t *)d_rows,
                                      d_vals_sorted, (size_t)total, 0, sort_bits, ctx->stream);
        if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_pairs failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(unpack_luggage_kernel, grid, dim3(256), 0, ctx->stream, d_keys_sorted, d_vals_sorted, total, band_bits, sort_bits,
                           d_keys_sorted, d_sorted_digests, d_sorted_rows);
    } else {
        const uint64_t *d_dig_in = d_dig;
        const int band_major = sig_dtype == kSigDigestsBM ? 1 : 0;
        if (sig_dtype == kSigDigests || sig_dtype == kSigDigestsBM) d_dig_in = static_cast<const uint64_t *>(d_sig);  // they are there already
        else if (int rc = launch_band_digests(ctx, d_sig, sig_dtype, n, k, bands, r, d_dig)) return rc;
        hipLaunchKernelGGL(band_keys_for_sort_kernel, grid, dim3(256), 0, ctx->stream, d_dig_in, n, bands, band_bits, sort_bits, band_major, d_keys, d_rows);
        MHX_HIP_CHECK(hipGetLastError());
        e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, (const uint64_t *)d_keys, d_keys_sorted, (const uint32_t *)d_rows,
                                      d_sorted_rows, (size_t)total, 0, sort_bits, ctx->stream);
        if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_pairs failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(gather_digests_kernel, grid, dim3(256), 0, ctx->stream, d_dig_in, d_sorted_rows, n, bands, total, band_major,
                           d_sorted_digests);
    }
    MHX_HIP_CHECK(hipMemsetAsync(d_mixed, 0, (size_t)total, ctx->stream));
    hipLaunchKernelGGL(mark_mixed_runs_kernel, grid, dim3(256), 0, ctx->stream, d_keys_sorted, d_sorted_digests, n, total, d_mixed);
    hipLaunchKernelGGL(order_mixed_runs_kernel, grid, dim3(256), 0, ctx->stream, d_keys_sorted, d_mixed, n, total,
                       d_sorted_digests, d_sorted_rows);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

int launch_lsh_query(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n, int32_t bands,
                     int32_t r, const void *d_q_sig, const void *d_idx_sig, int sig_dtype, int32_t k, int64_t m,
                     int64_t *d_pairs, int64_t capacity, int64_t *n_pairs) {
    *n_pairs = 0;
    const int64_t total = m * (int64_t)bands;
    if (total == 0 || n == 0) return MHX_OK;
    // scratch[4]: probe digests u64[total] | first u32[total] | count u32[total] | where u64[total] | scan temporary
    const size_t dig_bytes = ((sizeof(uint64_t) * (size_t)total) + 255) & ~(size_t)255;
    const size_t u32_bytes = ((sizeof(uint32_t) * (size_t)total) + 255) & ~(size_t)255;
    const size_t scan_tmp = scan_tmp_bytes(total);
    if (int rc = ctx->ensure_scratch(4, 2 * dig_bytes + 2 * u32_bytes + 256 + scan_tmp)) return rc;
    char *base = (char *)ctx->scratch[4];
    uint64_t *d_qdig = (uint64_t *)base;
    uint32_t *d_first = (uint32_t *)(base + dig_bytes);
    uint32_t *d_count = (uint32_t *)(base + dig_bytes + u32_bytes);
    uint64_t *d_where = (uint64_t *)(base + dig_bytes + 2 * u32_bytes);
    void *d_scan_tmp = base + 2 * dig_bytes + 2 * u32_bytes + 256;
    if (int rc = launch_band_digests(ctx, d_q_sig, sig_dtype, m, k, bands, r, d_qdig)) return rc;
    const dim3 grid(grid_for(ctx, total));
    hipLaunchKernelGGL(query_ranges_kernel, grid, dim3(256), 0, ctx->stream, d_qdig, m, bands, d_sorted_digests, n, d_first, d_count);
    MHX_HIP_CHECK(hipGetLastError());
    uint64_t *d_raw_total = nullptr;
    if (int rc = device_exclusive_scan(ctx, CountsIn{d_count}, WhereOut{d_where}, total, d_scan_tmp, &d_raw_total)) return rc;
    uint64_t raw_total = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&raw_total, d_raw_total, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const int64_t raw = (int64_t)raw_total;
    if (raw == 0) return MHX_OK;
    if ((size_t)raw * 16 > (size_t)ctx->hbm_bytes / 2)
        return fail(MHX_ERR_OOM, "%lld candidates before deduplication do not fit in device memory", (long long)raw);
    // scratch[3]: raw u64[raw] | sorted u64[raw] | count u64 | sort / select temporary
    const size_t raw_bytes = ((sizeof(uint64_t) * (size_t)raw) + 255) & ~(size_t)255;
    size_t sort_tmp = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, sort_tmp, (const uint64_t *)nullptr, (uint64_t *)nullptr, (size_t)raw, 0, 64, ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim size query failed: %s", hipGetErrorString(e));
    const size_t tmp_bytes = std::max(sort_tmp, scan_tmp_bytes(raw));
    if (int rc = ctx->ensure_scratch(3, 2 * raw_bytes + 256 + tmp_bytes)) return rc;
    uint64_t *d_raw = (uint64_t *)ctx->scratch[3];
    uint64_t *d_sorted = (uint64_t *)((char *)ctx->scratch[3] + raw_bytes);
    void *d_tmp = (char *)ctx->scratch[3] + 2 * raw_bytes + 256;
    const bool verify = d_idx_sig != nullptr;
    if (sig_dtype == MHX_U32) {
        if (verify)
            hipLaunchKernelGGL((query_emit_kernel<uint32_t, true>), grid, dim3(256), 0, ctx->stream, d_first, d_count, d_where, m, bands, n,
                               d_sorted_rows, (const uint32_t *)d_q_sig, (const uint32_t *)d_idx_sig, k, r, d_raw);
        else
            hipLaunchKernelGGL((query_emit_kernel<uint32_t, false>), grid, dim3(256), 0, ctx->stream, d_first, d_count, d_where, m, bands, n,
                               d_sorted_rows, (const uint32_t *)d_q_sig, (const uint32_t *)d_idx_sig, k, r, d_raw);
    } else {
        if (verify)
            hipLaunchKernelGGL((query_emit_kernel<uint64_t, true>), grid, dim3(256), 0, ctx->stream, d_first, d_count, d_where, m, bands, n,
                               d_sorted_rows, (const uint64_t *)d_q_sig, (const uint64_t *)d_idx_sig, k, r, d_raw);
        else
            hipLaunchKernelGGL((query_emit_kernel<uint64_t, false>), grid, dim3(256), 0, ctx->stream, d_first, d_count, d_where, m, bands, n,
                               d_sorted_rows, (const uint64_t *)d_q_sig, (const uint64_t *)d_idx_sig, k, r, d_raw);
    }
    MHX_HIP_CHECK(hipGetLastError());
    e = rocprim::radix_sort_keys(d_tmp, sort_tmp, (const uint64_t *)d_raw, d_sorted, (size_t)raw, 0, 64, ctx->stream);
    if (e != hipSuccess) return fail(MHX_ERR_HIP, "rocprim::radix_sort_keys failed: %s", hipGetErrorString(e));
    uint64_t *d_cnt = nullptr;  // unique: the run heads of the sorted candidates, packed
    if (int rc = device_exclusive_scan(ctx, HeadsIn{d_sorted}, CompactOut{d_sorted, d_raw}, raw, d_tmp, &d_cnt)) return rc;
    uint64_t unique_count = 0, last_key = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&unique_count, d_cnt, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (unique_count > 0) {  // a failed verification left ~0, which sorts last
        MHX_HIP_CHECK(hipMemcpyAsync(&last_key, d_raw + (unique_count - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (last_key == ~0ull) --unique_count;
    }
    *n_pairs = (int64_t)unique_count;
    if ((int64_t)unique_count > capacity || unique_count == 0) return MHX_OK;  // caller sees n_pairs > capacity and calls again
    hipLaunchKernelGGL(unpack_pairs_kernel, dim3(grid_for(ctx, (int64_t)unique_count)), dim3(256), 0, ctx->stream, d_raw,
                       (int64_t)unique_count, d_pairs);
    MHX_HIP_CHECK(hipGetLastError());
    return MHX_OK;
}

}  // namespace mhx

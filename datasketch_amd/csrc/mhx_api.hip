// mhx_api.hip -- C ABI of libmhx (include/mhx.h): context, device memory, events and the
// host-buffer entry points that stage through device scratch.  Product code: no oracle here.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "mhx_internal.h"

namespace mhx {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

void forgive() { g_last_error.clear(); }

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int bbit_slot_size(int b) {  // ref: datasketch/b_bit_minhash.py:147-160
    if (b == 1) return 1;
    if (b == 2) return 2;
    if (b <= 4) return 4;
    if (b <= 8) return 8;
    if (b <= 16) return 16;
    return 32;
}

// ---- device allocations of the library, all through here -------------------------------------------------------
// Normally hipMalloc / hipFree.  In guard mode (environment MHX_GUARD_ALLOC=<align>, or mhx_debug_guard_alloc) every
// allocation is mapped with the HIP virtual-memory API between two reserved, UNMAPPED granules and placed so that its
// last byte (align > 0; rounded up to `align` bytes) or its first byte (align < 0) abuts an unmapped page: a kernel that
// reads or writes past what it was given faults ("Memory access fault by GPU node ...") instead of silently touching a
// neighbour.  tests/test_guard_pages.py runs the GPU parity suite this way.
namespace {
struct GuardRec {
    void *va = nullptr;      // reserved range: [granule unmapped][mapped][granule unmapped]
    size_t reserved = 0, mapped = 0, granule = 0;
    hipMemGenericAllocationHandle_t handle{};
};
std::mutex g_guard_mu;
std::unordered_map<void *, GuardRec> g_guard;
int g_guard_align = -0x7fffffff;  // not read yet
unsigned long long g_guard_count = 0;

int guard_align() {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    if (g_guard_align == -0x7fffffff) {
        const char *e = getenv("MHX_GUARD_ALLOC");
        int a = e ? atoi(e) : 0;
        const int m = a < 0 ? -a : a;
        if (a != 0 && (m > 4096 || (m & (m - 1)))) {  // the rule of mhx_debug_guard_alloc: 0 or +-(a power of two <= 4096)
            fprintf(stderr, "libmhx: MHX_GUARD_ALLOC=%s is not 0 or +-(a power of two <= 4096): guard pages stay off\n", e);
            a = 0;
        }
        g_guard_align = a;
    }
    return g_guard_align;
}

hipError_t guard_malloc(void **p, size_t bytes, int align) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    if (gran == 0) return hipErrorNotSupported;
    const size_t a = (size_t)(align < 0 ? -align : align);
    const size_t want = (std::max<size_t>(bytes, 1) + a - 1) / a * a;
    GuardRec r;
    r.granule = gran;
    r.mapped = (want + gran - 1) / gran * gran;
    r.reserved = r.mapped + 2 * gran;
    e = hipMemAddressReserve(&r.va, r.reserved, gran, nullptr, 0);
    if (e != hipSuccess) return e;
    e = hipMemCreate(&r.handle, r.mapped, &prop, 0);
    if (e != hipSuccess) {
        (void)hipMemAddressFree(r.va, r.reserved);
        return e;
    }
    char *lo = static_cast<char *>(r.va) + gran;
    e = hipMemMap(lo, r.mapped, 0, r.handle, 0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(lo, r.mapped, &acc, 1);
        if (e != hipSuccess) (void)hipMemUnmap(lo, r.mapped);
    }
    if (e != hipSuccess) {
        (void)hipMemRelease(r.handle);
        (void)hipMemAddressFree(r.va, r.reserved);
        return e;
    }
    *p = align < 0 ? lo : lo + (r.mapped - want);
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard[*p] = r;
    ++g_guard_count;
    return hipSuccess;
}
}  // namespace

// caller: a block handed out by mhx_dev_alloc (placed with the guard alignment as it is); the library's own blocks --
// staging, tables, rocPRIM temporaries -- keep the 256-byte alignment hipMalloc gives them and that they are carved up by
// Poison mode (environment MHX_POISON_ALLOC=<byte 0..255>, or mhx_debug_poison_alloc): every fresh block is filled with
// that byte before it is handed out.  hipMalloc's memory is zero on a freshly booted board and whatever the previous
// tenant left on a used one; a kernel that reads a word nobody wrote works on the first and faults -- or answers
// wrongly -- on the second, box by box.  0xFF makes such a read a NaN, a -1 or a huge offset, every time.
static int g_poison = -0x7fffffff;  // not read yet
static int poison_byte() {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    if (g_poison == -0x7fffffff) {
        const char *e = getenv("MHX_POISON_ALLOC");
        g_poison = e ? (atoi(e) & 255) : -1;
    }
    return g_poison;
}

hipError_t dev_malloc(void **p, size_t bytes, bool caller) {
    int align = guard_align();
    hipError_t e;
    if (align == 0) {
        e = hipMalloc(p, bytes);
    } else {
        if (!caller) align = align < 0 ? std::min(align, -256) : std::max(align, 256);
        e = guard_malloc(p, bytes, align);
    }
    const int poison = poison_byte();
    if (e == hipSuccess && poison >= 0) {
        e = hipMemset(*p, poison, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
}

hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    GuardRec r;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        const auto it = g_guard.find(p);
        if (it == g_guard.end()) return hipFree(p);
        r = it->second;
        g_guard.erase(it);
    }
    // The physical pages go back, the address range stays reserved for the life of the process: a range handed out again
    // (hipMemAddressFree, then a new reservation at the same address) was read through STALE translations by the next
    // kernels on this driver -- whole inputs seen as zeros or as the previous tenant's bytes (measured: 172 of 407 guard
    // cases wrong with address reuse, none without; profiles/r04_guard_pages.txt).  A freed block thus stays an
    // unmapped hole, which is what a use-after-free should hit anyway.
    hipError_t e = hipDeviceSynchronize();
    char *lo = static_cast<char *>(r.va) + r.granule;
    const hipError_t e1 = hipMemUnmap(lo, r.mapped), e2 = hipMemRelease(r.handle);
    if (e == hipSuccess) e = e1;
    if (e == hipSuccess) e = e2;
    return e;
}

bool guard_mode() { return guard_align() != 0; }

}  // namespace mhx

using mhx::fail;

int mhx_ctx::activate() const {
    MHX_HIP_CHECK(hipSetDevice(device));
    return MHX_OK;
}

int mhx_ctx::ensure_scratch(int slot, size_t bytes) {
    // (guard mode: exactly what was asked for, every time, so that the end of the slot is the end of the mapping)
    const bool guard = mhx::guard_mode();
    if (guard ? (bytes == scratch_bytes[slot] && scratch[slot]) : bytes <= scratch_bytes[slot]) return MHX_OK;
    const size_t old = guard ? 0 : scratch_bytes[slot];
    if (scratch[slot]) {
        MHX_HIP_CHECK(hipStreamSynchronize(stream));
        MHX_HIP_CHECK(mhx::dev_free(scratch[slot]));
        scratch[slot] = nullptr;
        scratch_bytes[slot] = 0;
    }
    // grow geometrically so repeated slightly larger calls do not reallocate every time
    size_t want = std::max(bytes, old + old / 2);
    if (!guard) want = (want + 255) & ~(size_t)255;
    hipError_t e = mhx::dev_malloc(&scratch[slot], want);
    if (e != hipSuccess && want != bytes) {
        want = (bytes + 255) & ~(size_t)255;
        e = mhx::dev_malloc(&scratch[slot], want);
    }
    if (e != hipSuccess) {
        scratch[slot] = nullptr;
        return fail(MHX_ERR_OOM, "device scratch allocation of %zu bytes failed: %s", want,
                    hipGetErrorString(e));
    }
    scratch_bytes[slot] = want;
    return MHX_OK;
}

int mhx_ctx::ensure_copy_streams() {
    if (!copy_in) MHX_HIP_CHECK(hipStreamCreateWithFlags(&copy_in, hipStreamNonBlocking));
    if (!copy_out) MHX_HIP_CHECK(hipStreamCreateWithFlags(&copy_out, hipStreamNonBlocking));
    return MHX_OK;
}

int mhx_ctx::ensure_redo(int64_t n_sets) {
    if (n_sets <= redo_capacity && d_redo) return MHX_OK;
    if (d_redo) {
        (void)hipStreamSynchronize(stream);
        (void)mhx::dev_free(d_redo);
        d_redo = nullptr;
        redo_capacity = 0;
    }
    const int64_t cap = std::max<int64_t>(n_sets + n_sets / 4, 1024);
    hipError_t e = mhx::dev_malloc(reinterpret_cast<void **>(&d_redo), (size_t)cap + 64);
    if (e != hipSuccess) {
        d_redo = nullptr;
        return fail(MHX_ERR_OOM, "redo flag allocation of %lld bytes failed: %s", (long long)cap, hipGetErrorString(e));
    }
    redo_capacity = cap;
    return MHX_OK;
}

int mhx_ctx::ensure_work() {
    if (d_work) return MHX_OK;
    hipError_t e = mhx::dev_malloc(reinterpret_cast<void **>(&d_work), mhx::kWorkBytes);
    if (e != hipSuccess) {
        d_work = nullptr;
        return fail(MHX_ERR_OOM, "work counter allocation failed: %s", hipGetErrorString(e));
    }
    // on the kernels' own stream (created non-blocking: nothing orders the null stream before it); word 8, what the last call
    // learned about the corpus, lives across calls
    MHX_HIP_CHECK(hipMemsetAsync(d_work, 0, mhx::kWorkBytes, stream));
    return MHX_OK;
}

extern "C" {

const char *mhx_last_error(void) { return mhx::g_last_error.c_str(); }

const char *mhx_version(void) { return "mhx 0.1.0 (gfx950)"; }

int mhx_device_count(int *count) {
    if (!count) return fail(MHX_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return MHX_OK;  // "no device" is an answer, not an error (ref: minhash.py:38-48)
    }
    *count = n;
    return MHX_OK;
}

int mhx_ctx_create(int device, mhx_ctx **out) {
    if (!out) return fail(MHX_ERR_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(MHX_ERR_NO_DEVICE, "no HIP device is available");
    }
    if (device < 0 || device >= n) return fail(MHX_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    MHX_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MHX_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    mhx_ctx *ctx = new mhx_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    ctx->lds_per_block = (int64_t)prop.sharedMemPerBlock;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(MHX_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return MHX_OK;
}

int mhx_ctx_destroy(mhx_ctx *ctx) {
    if (!ctx) return MHX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 5; ++i)
        if (ctx->scratch[i]) (void)mhx::dev_free(ctx->scratch[i]);
    if (ctx->d_stats) (void)mhx::dev_free(ctx->d_stats);
    if (ctx->d_redo) (void)mhx::dev_free(ctx->d_redo);
    if (ctx->d_work) (void)mhx::dev_free(ctx->d_work);
    if (ctx->copy_in) (void)hipStreamDestroy(ctx->copy_in);
    if (ctx->copy_out) (void)hipStreamDestroy(ctx->copy_out);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MHX_OK;
}

int mhx_ctx_synchronize(mhx_ctx *ctx) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_ctx_release_scratch(mhx_ctx *ctx) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 5; ++i) {
        if (ctx->scratch[i]) MHX_HIP_CHECK(mhx::dev_free(ctx->scratch[i]));
        ctx->scratch[i] = nullptr;
        ctx->scratch_bytes[i] = 0;
    }
    if (ctx->d_redo) {
        MHX_HIP_CHECK(mhx::dev_free(ctx->d_redo));
        ctx->d_redo = nullptr;
        ctx->redo_capacity = 0;
    }
    return MHX_OK;
}

int mhx_ctx_device_info(mhx_ctx *ctx, char *name, int name_len, int *cus, int64_t *hbm_bytes) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (name && name_len > 0) {
        strncpy(name, ctx->name, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cus) *cus = ctx->num_cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return MHX_OK;
}

int mhx_ctx_set_option(mhx_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return fail(MHX_ERR_INVALID, "ctx/key is NULL");
    MHX_GUARD(ctx);
    if (!strcmp(key, "minhash.path")) ctx->opt_minhash_path = value;
    else if (!strcmp(key, "minhash.split")) ctx->opt_minhash_split = value;
    else if (!strcmp(key, "minhash.packed")) ctx->opt_minhash_packed = value;
    else if (!strcmp(key, "minhash.ties")) ctx->opt_minhash_ties = value;
    else if (!strcmp(key, "minhash.p3")) ctx->opt_minhash_p3 = value;
    else if (!strcmp(key, "minhash.share")) ctx->opt_minhash_share = value;
    else if (!strcmp(key, "minhash.adapt")) ctx->opt_minhash_adapt = value;
    else if (!strcmp(key, "blocks_per_cu")) ctx->opt_blocks_per_cu = value;
    else if (!strcmp(key, "minhash.alias")) ctx->opt_minhash_alias = value;
    else if (!strcmp(key, "minhash.prefetch")) ctx->opt_minhash_prefetch = value;
    else if (!strcmp(key, "weighted.path")) ctx->opt_weighted_path = value;
    else if (!strcmp(key, "weighted.direct")) ctx->opt_weighted_direct = value;
    else if (!strcmp(key, "weighted.split")) ctx->opt_weighted_split = value;
    else if (!strcmp(key, "weighted.tail")) ctx->opt_weighted_tail = value;
    else if (!strcmp(key, "weighted.debug")) ctx->opt_weighted_debug = value;
    else if (!strcmp(key, "weighted.kernel")) ctx->opt_weighted_kernel = value;
    else if (!strcmp(key, "weighted.plan")) ctx->opt_weighted_plan = value;
    else if (!strcmp(key, "weighted.rescue")) ctx->opt_weighted_rescue = value;
    else if (!strcmp(key, "weighted.min_dim")) ctx->opt_weighted_min_dim = value;
    else if (!strcmp(key, "host.chunk_bytes")) ctx->opt_host_chunk_bytes = value;
    else if (!strcmp(key, "lsh.sort_bits")) ctx->opt_lsh_sort_bits = value;
    else if (!strcmp(key, "lsh.gather")) ctx->opt_lsh_gather = value;
    else if (!strcmp(key, "lsh.sort")) ctx->opt_lsh_sort = value;
    else if (!strcmp(key, "lsh.levels")) ctx->opt_lsh_levels = value;
    else if (!strcmp(key, "lsh.chunk")) ctx->opt_lsh_chunk = value;
    else if (!strcmp(key, "lsh.team")) ctx->opt_lsh_team = value;
    else if (!strcmp(key, "lsh.bigbins")) ctx->opt_lsh_bigbins = value;
    else if (!strcmp(key, "pack.fused")) ctx->opt_pack_fused = value;
    else if (!strcmp(key, "weighted.refill")) ctx->opt_weighted_refill = value;
    else if (!strcmp(key, "lsh.prehash")) ctx->opt_lsh_prehash = value;
    else return fail(MHX_ERR_INVALID, "unknown option '%s'", key);
    return MHX_OK;
}

// What the previous MinHash call on this context learned about the corpus (d_work word 8, written by the last launch of every
// call and read by the first launch of the next one): 0 = one-candidate proof first, 1 = most sets defeat it (tie-tolerant proof
// first), 2 = heavily repeated tokens.  Timings depend on it, results never; reset = 1 puts it back to 0 (a fresh context).
int mhx_ctx_minhash_mode(mhx_ctx *ctx, int reset, int *mode) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    if (int rc = ctx->ensure_work()) return rc;
    unsigned int word = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&word, ctx->d_work + 8, sizeof(word), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (mode) *mode = (int)word;
    if (reset) MHX_HIP_CHECK(hipMemsetAsync(ctx->d_work + 8, 0, sizeof(word), ctx->stream));
    return MHX_OK;
}

// Which sets of the last MinHash call left the fast path (see mhx.h).  The flags are the launches' own hand-over bytes: the sieve
// launch writes 0 / 1 for every set, the second launch turns the 1 of a set it could not certify either into 2.
int mhx_ctx_minhash_flags(mhx_ctx *ctx, int64_t n_sets, uint8_t *flags) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    MHX_REQUIRE(n_sets >= 0 && (flags || n_sets == 0), "NULL flags");
    if (ctx->redo_sets != n_sets || (n_sets > 0 && !ctx->d_redo))
        return fail(MHX_ERR_INVALID, "the last MinHash call on this context kept flags for %lld sets, not %lld (one huge set split over waves and "
                    "minhash.path != 0 keep none)", (long long)ctx->redo_sets, (long long)n_sets);
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n_sets > 0) MHX_HIP_CHECK(hipMemcpy(flags, ctx->d_redo, (size_t)n_sets, hipMemcpyDeviceToHost));
    return MHX_OK;
}

int mhx_ctx_counters(mhx_ctx *ctx, int enable, uint64_t out[MHX_NUM_COUNTERS]) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (out) {
        for (int i = 0; i < MHX_NUM_COUNTERS; ++i) out[i] = 0;
        if (ctx->d_stats)
            MHX_HIP_CHECK(hipMemcpy(out, ctx->d_stats, sizeof(uint64_t) * MHX_NUM_COUNTERS, hipMemcpyDeviceToHost));
    }
    if (enable && !ctx->d_stats) {
        hipError_t e = mhx::dev_malloc(reinterpret_cast<void **>(&ctx->d_stats), sizeof(uint64_t) * MHX_NUM_COUNTERS);
        if (e != hipSuccess) {
            ctx->d_stats = nullptr;
            return fail(MHX_ERR_OOM, "counter allocation failed: %s", hipGetErrorString(e));
        }
    }
    if (!enable && ctx->d_stats) {
        MHX_HIP_CHECK(mhx::dev_free(ctx->d_stats));
        ctx->d_stats = nullptr;
    }
    if (ctx->d_stats) MHX_HIP_CHECK(hipMemset(ctx->d_stats, 0, sizeof(uint64_t) * MHX_NUM_COUNTERS));
    return MHX_OK;
}

// ---- device memory -------------------------------------------------------------------------
int mhx_debug_guard_alloc(int align, int64_t *granule, int64_t *live) {
    if (align != 0 && (align < -4096 || align > 4096 || ((align < 0 ? -align : align) & ((align < 0 ? -align : align) - 1))))
        return fail(MHX_ERR_INVALID, "guard alignment must be 0 (off) or +-(a power of two <= 4096), got %d", align);
    (void)mhx::guard_mode();  // read the environment first: this call overrides it
    {
        std::lock_guard<std::mutex> lk(mhx::g_guard_mu);
        mhx::g_guard_align = align;
        if (live) *live = (int64_t)mhx::g_guard.size();
    }
    if (granule) {
        *granule = 0;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            return align ? fail(MHX_ERR_NO_DEVICE, "no HIP device is available") : MHX_OK;
        }
        int device = 0;
        MHX_HIP_CHECK(hipGetDevice(&device));
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t g = 0;
        hipError_t e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(MHX_ERR_UNSUPPORTED, "hipMemGetAllocationGranularity failed: %s", hipGetErrorString(e));
        }
        *granule = (int64_t)g;
    }
    return MHX_OK;
}

int mhx_debug_poison_alloc(int byte_value) {
    if (byte_value < -1 || byte_value > 255) return fail(MHX_ERR_INVALID, "poison byte must be -1 (off) or 0..255, got %d", byte_value);
    (void)mhx::poison_byte();  // read the environment first: this call overrides it
    std::lock_guard<std::mutex> lk(mhx::g_guard_mu);
    mhx::g_poison = byte_value;
    return MHX_OK;
}

int mhx_dev_alloc(mhx_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return fail(MHX_ERR_INVALID, "ctx/dptr is NULL");
    MHX_GUARD(ctx);
    *dptr = nullptr;
    if (int rc = ctx->activate()) return rc;
    hipError_t e = mhx::dev_malloc(dptr, bytes ? bytes : 1, true);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(MHX_ERR_OOM, "device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    }
    return MHX_OK;
}

int mhx_dev_free(mhx_ctx *ctx, void *dptr) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (!dptr) return MHX_OK;
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MHX_HIP_CHECK(mhx::dev_free(dptr));
    return MHX_OK;
}

int mhx_host_alloc(mhx_ctx *ctx, size_t bytes, void **ptr) {
    if (!ctx || !ptr) return fail(MHX_ERR_INVALID, "ctx/ptr is NULL");
    MHX_GUARD(ctx);
    *ptr = nullptr;
    if (int rc = ctx->activate()) return rc;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(MHX_ERR_OOM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    return MHX_OK;
}

int mhx_host_free(mhx_ctx *ctx, void *ptr) {
    if (!ptr) return MHX_OK;
    if (!ctx) {  // the context that allocated it is gone (and with it every transfer that could still use the block)
        MHX_HIP_CHECK(hipHostFree(ptr));
        return MHX_OK;
    }
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->copy_in) MHX_HIP_CHECK(hipStreamSynchronize(ctx->copy_in));
    MHX_HIP_CHECK(hipHostFree(ptr));
    return MHX_OK;
}

int mhx_memcpy_h2d(mhx_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (!bytes) return MHX_OK;
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_memcpy_d2h(mhx_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (!bytes) return MHX_OK;
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_memcpy_d2d(mhx_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (!bytes) return MHX_OK;
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MHX_OK;
}

int mhx_memset_dev(mhx_ctx *ctx, void *dst, int byte_value, size_t bytes) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    if (!bytes) return MHX_OK;
    if (int rc = ctx->activate()) return rc;
    MHX_HIP_CHECK(hipMemsetAsync(dst, byte_value, bytes, ctx->stream));
    return MHX_OK;
}

// ---- events --------------------------------------------------------------------------------
int mhx_event_create(mhx_ctx *ctx, mhx_event **ev) {
    if (!ctx || !ev) return fail(MHX_ERR_INVALID, "ctx/ev is NULL");
    MHX_GUARD(ctx);
    if (int rc = ctx->activate()) return rc;
    mhx_event *e = new mhx_event();
    e->ctx = ctx;
    hipError_t err = hipEventCreate(&e->ev);
    if (err != hipSuccess) {
        delete e;
        return fail(MHX_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(err));
    }
    *ev = e;
    return MHX_OK;
}

int mhx_event_record(mhx_event *ev) {
    if (!ev) return fail(MHX_ERR_INVALID, "event is NULL");
    MHX_GUARD(ev->ctx);
    MHX_HIP_CHECK(hipEventRecord(ev->ev, ev->ctx->stream));
    return MHX_OK;
}

int mhx_event_synchronize(mhx_event *ev) {
    if (!ev) return fail(MHX_ERR_INVALID, "event is NULL");
    MHX_HIP_CHECK(hipEventSynchronize(ev->ev));
    return MHX_OK;
}

int mhx_event_elapsed_ms(mhx_event *start, mhx_event *stop, float *ms) {
    if (!start || !stop || !ms) return fail(MHX_ERR_INVALID, "event/ms is NULL");
    MHX_HIP_CHECK(hipEventElapsedTime(ms, start->ev, stop->ev));
    return MHX_OK;
}

int mhx_event_destroy(mhx_event *ev) {
    if (!ev) return MHX_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return MHX_OK;
}

// ---- MinHash -------------------------------------------------------------------------------
int mhx_perm_create(mhx_ctx *ctx, const uint64_t *a, const uint64_t *b, int32_t num_perm,
                    mhx_perm **out) {
    if (!ctx || !a || !b || !out) return fail(MHX_ERR_INVALID, "NULL argument");
    MHX_GUARD(ctx);
    MHX_REQUIRE(num_perm > 0, "num_perm must be positive, got %d", num_perm);
    if (int rc = ctx->activate()) return rc;
    mhx_perm *p = new mhx_perm();
    p->ctx = ctx;
    p->num_perm = num_perm;
    const size_t bytes = sizeof(uint64_t) * (size_t)num_perm;
    hipError_t e = mhx::dev_malloc((void **)&p->d_a, 2 * bytes);
    if (e != hipSuccess) {
        delete p;
        return fail(MHX_ERR_OOM, "hipMalloc for permutations failed: %s", hipGetErrorString(e));
    }
    p->d_b = p->d_a + num_perm;
    e = hipMemcpyAsync(p->d_a, a, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_b, b, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)mhx::dev_free(p->d_a);
        delete p;
        return fail(MHX_ERR_HIP, "uploading permutations failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return MHX_OK;
}

int mhx_perm_destroy(mhx_perm *perm) {
    if (!perm) return MHX_OK;
    MHX_GUARD(perm->ctx);
    (void)hipSetDevice(perm->ctx->device);
    (void)hipStreamSynchronize(perm->ctx->stream);
    (void)mhx::dev_free(perm->d_a);
    delete perm;
    return MHX_OK;
}

int mhx_minhash_bulk_dev(mhx_perm *perm, const void *d_hv, int hv_dtype, const int64_t *d_offsets,
                         int64_t fixed_len, int64_t n_sets, int64_t total_tokens,
                         const uint64_t *d_init, int64_t init_stride, void *d_out, int out_dtype) {
    if (!perm) return fail(MHX_ERR_INVALID, "perm is NULL");
    MHX_GUARD(perm->ctx);
    MHX_REQUIRE(n_sets >= 0, "n_sets must be >= 0");
    MHX_REQUIRE(hv_dtype == MHX_U64 || hv_dtype == MHX_U32, "bad hv_dtype %d", hv_dtype);
    MHX_REQUIRE(out_dtype == MHX_U64 || out_dtype == MHX_U32, "bad out_dtype %d", out_dtype);
    MHX_REQUIRE(d_offsets || fixed_len >= 0, "fixed_len must be >= 0 when offsets is NULL");
    MHX_REQUIRE(init_stride == 0 || init_stride >= perm->num_perm, "init_stride must be 0 or >= num_perm");
    MHX_REQUIRE(total_tokens >= 0, "total_tokens must be >= 0");
    if (n_sets == 0) return MHX_OK;
    MHX_REQUIRE(d_out, "d_out is NULL");
    MHX_REQUIRE(d_hv || total_tokens == 0, "d_hv is NULL");
    if (int rc = perm->ctx->activate()) return rc;
    return mhx::launch_minhash_bulk(perm, d_hv, hv_dtype, d_offsets, fixed_len, n_sets, total_tokens,
                                    d_init, init_stride, d_out, out_dtype);
}

extern "C++" {
namespace {

constexpr int kNoSecondThread = -1000;  // internal: bulk_pipelined could not start its download thread

// One piece of a pipelined host call: sets [s0, s1) whose tokens are hv[t0, t1).
struct Piece {
    int64_t s0, s1, t0, t1;
};

// Cut the corpus into pieces of about `target` bytes (tokens in + signature rows out); a piece is
// at least one set, so one enormous set still becomes one piece.
std::vector<Piece> cut_pieces(const int64_t *offsets, int64_t fixed_len, int64_t n_sets, int64_t k, int64_t target,
                              int64_t tok_size = 8, int64_t out_size = 8) {
    std::vector<Piece> pieces;
    int64_t s0 = 0;
    while (s0 < n_sets) {
        int64_t s1;
        if (offsets) {
            // largest s1 with tok_size*(offsets[s1]-offsets[s0]) + out_size*k*(s1-s0) <= target: the cost is increasing in s1
            int64_t lo = s0 + 1, hi = n_sets;
            while (lo < hi) {
                const int64_t mid = lo + (hi - lo + 1) / 2;
                const int64_t cost = tok_size * (offsets[mid] - offsets[s0]) + out_size * k * (mid - s0);
                if (cost <= target) lo = mid; else hi = mid - 1;
            }
            s1 = lo;
        } else {
            const int64_t per_set = tok_size * fixed_len + out_size * k;
            s1 = std::min(n_sets, s0 + std::max<int64_t>(1, target / per_set));
        }
        Piece p;
        p.s0 = s0;
        p.s1 = s1;
        p.t0 = offsets ? offsets[s0] : s0 * fixed_len;
        p.t1 = offsets ? offsets[s1] : s1 * fixed_len;
        pieces.push_back(p);
        s0 = s1;
    }
    return pieces;
}

// Host corpus -> host signatures with the three legs overlapped: this thread uploads piece i+1
// (copy_in stream) while the kernels of piece i run (ctx->stream) and a second thread downloads the
// rows of piece i-1 (copy_out stream).  PCIe is full duplex, so a large call costs about
// max(upload, download) instead of their sum.  Device buffers hold the whole corpus (no reuse
// hazards); offsets stay absolute, so a piece is just a window of sets.
int bulk_pipelined(mhx_perm *perm, const char *hv, int hv_dtype, const int64_t *offsets, int64_t fixed_len, int64_t n_sets,
                   const uint64_t *init, int64_t init_stride, char *out, int out_dtype, char *d_hv, int64_t *d_off,
                   uint64_t *d_init, char *d_out, const std::vector<Piece> &pieces) {
    mhx_ctx *ctx = perm->ctx;
    const int64_t k = perm->num_perm;
    const size_t ts = hv_dtype == MHX_U32 ? 4 : 8, os = out_dtype == MHX_U32 ? 4 : 8;  // element sizes
    if (int rc = ctx->ensure_copy_streams()) return rc;
    const size_t n_pieces = pieces.size();
    std::vector<hipEvent_t> uploaded(n_pieces, nullptr), computed(n_pieces, nullptr);
    auto destroy_events = [&]() {
        for (hipEvent_t e : uploaded) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : computed) if (e) (void)hipEventDestroy(e);
    };
    for (size_t i = 0; i < n_pieces; ++i) {
        hipError_t e = hipEventCreateWithFlags(&uploaded[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&computed[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            destroy_events();
            return fail(MHX_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
    }
    // the download thread may only wait on an event after this thread has recorded it
    std::mutex mu;
    std::condition_variable cv;
    size_t recorded = 0;
    bool stop = false;
    hipError_t down_err = hipSuccess;
    auto download = [&]() {
        hipError_t e = hipSetDevice(ctx->device);
        for (size_t i = 0; i < n_pieces && e == hipSuccess; ++i) {
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return recorded > i || stop; });
                if (recorded <= i) break;  // stopped before this piece was launched
            }
            const Piece &p = pieces[i];
            e = hipStreamWaitEvent(ctx->copy_out, computed[i], 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync(out + os * (size_t)(p.s0 * k), d_out + os * (size_t)(p.s0 * k), os * (size_t)((p.s1 - p.s0) * k),
                                   hipMemcpyDeviceToHost, ctx->copy_out);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_out);
        down_err = e;
    };
    std::thread downloader;
    try {
        downloader = std::thread(download);
    } catch (const std::system_error &) {  // no thread to be had: the caller runs the call in one piece
        destroy_events();
        return kNoSecondThread;
    }
    auto finish = [&](int rc) -> int {
        {
            std::lock_guard<std::mutex> lock(mu);
            stop = true;
        }
        cv.notify_all();
        downloader.join();
        (void)hipStreamSynchronize(ctx->copy_in);
        (void)hipStreamSynchronize(ctx->stream);
        destroy_events();
        if (rc) return rc;
        if (down_err != hipSuccess) return fail(MHX_ERR_HIP, "downloading signatures failed: %s", hipGetErrorString(down_err));
        return MHX_OK;
    };
    for (size_t i = 0; i < n_pieces; ++i) {  // offsets stay absolute: hv[t] sits at d_hv[t]
        const Piece &p = pieces[i];
        hipError_t e = hipSuccess;
        if (p.t1 > p.t0)
            e = hipMemcpyAsync(d_hv + ts * (size_t)p.t0, hv + ts * (size_t)p.t0, ts * (size_t)(p.t1 - p.t0), hipMemcpyHostToDevice,
                               ctx->copy_in);
        if (e == hipSuccess && init && init_stride)
            e = hipMemcpyAsync(d_init + p.s0 * init_stride, init + p.s0 * init_stride,
                               sizeof(uint64_t) * (size_t)((p.s1 - p.s0) * init_stride), hipMemcpyHostToDevice, ctx->copy_in);
        if (e == hipSuccess) e = hipEventRecord(uploaded[i], ctx->copy_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, uploaded[i], 0);
        if (e != hipSuccess) return finish(fail(MHX_ERR_HIP, "uploading tokens failed: %s", hipGetErrorString(e)));
        const char *piece_hv = offsets ? d_hv : d_hv + ts * (size_t)p.t0;
        const int64_t *piece_off = offsets ? d_off + p.s0 : nullptr;
        const uint64_t *piece_init = !init ? nullptr : (init_stride ? d_init + p.s0 * init_stride : d_init);
        const int64_t first = offsets ? p.t0 : 0, last = offsets ? p.t1 : p.t1 - p.t0;
        if (int rc = mhx::launch_minhash_bulk(perm, piece_hv, hv_dtype, piece_off, fixed_len, p.s1 - p.s0, last,
                                              piece_init, init_stride, d_out + os * (size_t)(p.s0 * k), out_dtype, first))
            return finish(rc);
        e = hipEventRecord(computed[i], ctx->stream);
        if (e != hipSuccess) return finish(fail(MHX_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e)));
        {
            std::lock_guard<std::mutex> lock(mu);
            recorded = i + 1;
        }
        cv.notify_all();
    }
    return finish(MHX_OK);
}

}  // namespace
}  // extern "C++"

int mhx_minhash_bulk_typed(mhx_perm *perm, const void *hv, int hv_dtype, const int64_t *offsets, int64_t fixed_len,
                           int64_t n_sets, const uint64_t *init, int64_t init_stride, void *out, int out_dtype) {
    if (!perm) return fail(MHX_ERR_INVALID, "perm is NULL");
    MHX_GUARD(perm->ctx);
    MHX_REQUIRE(n_sets >= 0, "n_sets must be >= 0");
    MHX_REQUIRE(hv_dtype == MHX_U64 || hv_dtype == MHX_U32, "bad hv_dtype %d", hv_dtype);
    MHX_REQUIRE(out_dtype == MHX_U64 || out_dtype == MHX_U32, "bad out_dtype %d", out_dtype);
    if (n_sets == 0) return MHX_OK;
    MHX_REQUIRE(out, "out is NULL");
    MHX_REQUIRE(offsets || fixed_len >= 0, "fixed_len must be >= 0 when offsets is NULL");
    mhx_ctx *ctx = perm->ctx;
    if (int rc = ctx->activate()) return rc;
    const int64_t k = perm->num_perm;
    const size_t ts = hv_dtype == MHX_U32 ? 4 : 8, os = out_dtype == MHX_U32 ? 4 : 8;
    int64_t total = 0;
    if (offsets) {
        MHX_REQUIRE(offsets[0] >= 0, "offsets[0] must be >= 0");
        for (int64_t i = 0; i < n_sets; ++i)
            MHX_REQUIRE(offsets[i + 1] >= offsets[i], "offsets must be non-decreasing (row %lld)", (long long)i);
        total = offsets[n_sets];
    } else {
        total = n_sets * fixed_len;
    }
    MHX_REQUIRE(hv || total == 0, "hv is NULL");
    const size_t hv_bytes = ts * (size_t)total;
    const size_t off_bytes = offsets ? sizeof(int64_t) * (size_t)(n_sets + 1) : 0;
    const size_t out_bytes = os * (size_t)(n_sets * k);
    const size_t init_bytes = init ? sizeof(uint64_t) * (size_t)(init_stride ? n_sets * init_stride : k) : 0;
    if (int rc = ctx->ensure_scratch(0, hv_bytes + 256)) return rc;
    if (int rc = ctx->ensure_scratch(1, off_bytes + init_bytes + 512)) return rc;
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    char *d_hv = (char *)ctx->scratch[0];
    int64_t *d_off = offsets ? (int64_t *)ctx->scratch[1] : nullptr;
    uint64_t *d_init = init ? (uint64_t *)((char *)ctx->scratch[1] + ((off_bytes + 255) & ~(size_t)255)) : nullptr;
    char *d_out = (char *)ctx->scratch[2];

    // large corpora: upload, kernels and download overlap piece by piece
    const int64_t chunk_opt = ctx->opt_host_chunk_bytes;
    const int64_t target = chunk_opt > 0 ? chunk_opt : (int64_t)96 << 20;
    const bool pipelined = chunk_opt > 0 || (chunk_opt == 0 && hv_bytes + out_bytes > ((size_t)256 << 20));
    if (pipelined) {
        const std::vector<Piece> pieces = cut_pieces(offsets, fixed_len, n_sets, k, target, (int64_t)ts, (int64_t)os);
        if (pieces.size() > 1) {
            // small operands first, on the compute stream: every piece's kernels are ordered after them
            if (off_bytes) MHX_HIP_CHECK(hipMemcpyAsync(d_off, offsets, off_bytes, hipMemcpyHostToDevice, ctx->stream));
            if (init && !init_stride)
                MHX_HIP_CHECK(hipMemcpyAsync(d_init, init, init_bytes, hipMemcpyHostToDevice, ctx->stream));
            const int rc = bulk_pipelined(perm, (const char *)hv, hv_dtype, offsets, fixed_len, n_sets, init, init_stride,
                                          (char *)out, out_dtype, d_hv, d_off, d_init, d_out, pieces);
            if (rc != kNoSecondThread) return rc;
        }
    }
    if (hv_bytes) MHX_HIP_CHECK(hipMemcpyAsync(d_hv, hv, hv_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (off_bytes) MHX_HIP_CHECK(hipMemcpyAsync(d_off, offsets, off_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (init_bytes) MHX_HIP_CHECK(hipMemcpyAsync(d_init, init, init_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_minhash_bulk(perm, d_hv, hv_dtype, d_off, fixed_len, n_sets, total, d_init,
                                          init_stride, d_out, out_dtype))
        return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_minhash_bulk(mhx_perm *perm, const uint64_t *hv, const int64_t *offsets, int64_t fixed_len,
                     int64_t n_sets, const uint64_t *init, int64_t init_stride, uint64_t *out) {
    return mhx_minhash_bulk_typed(perm, hv, MHX_U64, offsets, fixed_len, n_sets, init, init_stride, out, MHX_U64);
}

// ---- token hashing (sha1_hash32 / sha1_hash64 of byte tokens) ----------------------------------
int mhx_sha1_tokens_dev(mhx_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_byte_offsets, int64_t n_tokens,
                        int out_dtype, void *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_tokens >= 0, "n_tokens must be >= 0");
    MHX_REQUIRE(out_dtype == MHX_U32 || out_dtype == MHX_U64, "out_dtype must be MHX_U32 or MHX_U64");
    if (n_tokens == 0) return MHX_OK;
    MHX_REQUIRE(d_byte_offsets && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_sha1_tokens(ctx, d_bytes, d_byte_offsets, n_tokens, out_dtype, d_out);
}

namespace {
// validate + upload a packed byte corpus: bytes -> scratch[0], byte offsets -> scratch[3]
int upload_tokens(mhx_ctx *ctx, const uint8_t *bytes, const int64_t *byte_offsets, int64_t n_tokens,
                  uint8_t **d_bytes, int64_t **d_offs) {
    MHX_REQUIRE(byte_offsets, "byte_offsets is NULL");
    MHX_REQUIRE(byte_offsets[0] == 0, "byte_offsets[0] must be 0");
    for (int64_t i = 0; i < n_tokens; ++i)
        MHX_REQUIRE(byte_offsets[i + 1] >= byte_offsets[i], "byte_offsets must be non-decreasing (token %lld)", (long long)i);
    const int64_t total = byte_offsets[n_tokens];
    MHX_REQUIRE(bytes || total == 0, "bytes is NULL");
    if (int rc = ctx->ensure_scratch(0, (size_t)total + 256)) return rc;
    if (int rc = ctx->ensure_scratch(3, sizeof(int64_t) * (size_t)(n_tokens + 1))) return rc;
    *d_bytes = (uint8_t *)ctx->scratch[0];
    *d_offs = (int64_t *)ctx->scratch[3];
    if (total) MHX_HIP_CHECK(hipMemcpyAsync(*d_bytes, bytes, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(*d_offs, byte_offsets, sizeof(int64_t) * (size_t)(n_tokens + 1), hipMemcpyHostToDevice,
                                 ctx->stream));
    return MHX_OK;
}
}  // namespace

int mhx_sha1_tokens(mhx_ctx *ctx, const uint8_t *bytes, const int64_t *byte_offsets, int64_t n_tokens,
                    int out_dtype, void *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_tokens >= 0, "n_tokens must be >= 0");
    MHX_REQUIRE(out_dtype == MHX_U32 || out_dtype == MHX_U64, "out_dtype must be MHX_U32 or MHX_U64");
    if (n_tokens == 0) return MHX_OK;
    MHX_REQUIRE(out, "out is NULL");
    if (int rc = ctx->activate()) return rc;
    uint8_t *d_bytes = nullptr;
    int64_t *d_offs = nullptr;
    if (int rc = upload_tokens(ctx, bytes, byte_offsets, n_tokens, &d_bytes, &d_offs)) return rc;
    const size_t out_bytes = (size_t)n_tokens * (out_dtype == MHX_U32 ? 4 : 8);
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    if (int rc = mhx::launch_sha1_tokens(ctx, d_bytes, d_offs, n_tokens, out_dtype, ctx->scratch[2])) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, ctx->scratch[2], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_minhash_bulk_bytes(mhx_perm *perm, const uint8_t *bytes, const int64_t *byte_offsets, int64_t n_tokens,
                           const int64_t *set_offsets, int64_t n_sets, const uint64_t *init, int64_t init_stride,
                           uint64_t *out) {
    return mhx_minhash_bulk_bytes_typed(perm, bytes, byte_offsets, n_tokens, MHX_U32, set_offsets, n_sets, init, init_stride, out);
}

int mhx_minhash_bulk_bytes_typed(mhx_perm *perm, const uint8_t *bytes, const int64_t *byte_offsets, int64_t n_tokens,
                                 int hash_dtype, const int64_t *set_offsets, int64_t n_sets, const uint64_t *init,
                                 int64_t init_stride, uint64_t *out) {
    if (!perm) return fail(MHX_ERR_INVALID, "perm is NULL");
    MHX_GUARD(perm->ctx);
    MHX_REQUIRE(hash_dtype == MHX_U32 || hash_dtype == MHX_U64, "hash_dtype must be MHX_U32 (sha1_hash32) or MHX_U64 (sha1_hash64)");
    MHX_REQUIRE(n_sets >= 0 && n_tokens >= 0, "n_sets and n_tokens must be >= 0");
    if (n_sets == 0) return MHX_OK;
    MHX_REQUIRE(out && set_offsets, "out/set_offsets is NULL");
    MHX_REQUIRE(set_offsets[0] == 0 && set_offsets[n_sets] == n_tokens, "set_offsets must run from 0 to n_tokens");
    for (int64_t i = 0; i < n_sets; ++i)
        MHX_REQUIRE(set_offsets[i + 1] >= set_offsets[i], "set_offsets must be non-decreasing (set %lld)", (long long)i);
    mhx_ctx *ctx = perm->ctx;
    if (int rc = ctx->activate()) return rc;
    const int64_t k = perm->num_perm;
    uint8_t *d_bytes = nullptr;
    int64_t *d_boffs = nullptr;
    if (n_tokens > 0)
        if (int rc = upload_tokens(ctx, bytes, byte_offsets, n_tokens, &d_bytes, &d_boffs)) return rc;
    // scratch[1]: set offsets | init | token hashes (uint32 or uint64);  scratch[2]: signatures
    const size_t hs = hash_dtype == MHX_U32 ? 4 : 8;
    const size_t off_bytes = sizeof(int64_t) * (size_t)(n_sets + 1);
    const size_t init_bytes = init ? sizeof(uint64_t) * (size_t)(init_stride ? n_sets * init_stride : k) : 0;
    const size_t off_pad = (off_bytes + 255) & ~(size_t)255, init_pad = (init_bytes + 255) & ~(size_t)255;
    const size_t out_bytes = sizeof(uint64_t) * (size_t)(n_sets * k);
    if (int rc = ctx->ensure_scratch(1, off_pad + init_pad + hs * (size_t)n_tokens + 256)) return rc;
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    int64_t *d_soffs = (int64_t *)ctx->scratch[1];
    uint64_t *d_init = init ? (uint64_t *)((char *)ctx->scratch[1] + off_pad) : nullptr;
    void *d_hv = (char *)ctx->scratch[1] + off_pad + init_pad;
    uint64_t *d_out = (uint64_t *)ctx->scratch[2];
    MHX_HIP_CHECK(hipMemcpyAsync(d_soffs, set_offsets, off_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (init_bytes) MHX_HIP_CHECK(hipMemcpyAsync(d_init, init, init_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_sha1_tokens(ctx, d_bytes, d_boffs, n_tokens, hash_dtype, d_hv)) return rc;
    if (int rc = mhx::launch_minhash_bulk(perm, d_hv, hash_dtype, d_soffs, 0, n_sets, n_tokens, d_init, init_stride, d_out,
                                          MHX_U64))
        return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_minhash_update_batch(mhx_perm *perm, const uint64_t *hv, int64_t n, uint64_t *hashvalues) {
    if (!perm) return fail(MHX_ERR_INVALID, "perm is NULL");
    MHX_GUARD(perm->ctx);
    MHX_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return MHX_OK;  // ref: minhash.py:265-266
    MHX_REQUIRE(hv && hashvalues, "hv/hashvalues is NULL");
    return mhx_minhash_bulk(perm, hv, nullptr, n, 1, hashvalues, 0, hashvalues);
}

int mhx_minhash_merge_dev(mhx_ctx *ctx, const uint64_t *d_x, const uint64_t *d_y, int64_t count,
                          uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(count >= 0, "count must be >= 0");
    if (count == 0) return MHX_OK;
    MHX_REQUIRE(d_x && d_y && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_minhash_merge(ctx, d_x, d_y, count, d_out);
}

int mhx_minhash_merge(mhx_ctx *ctx, const uint64_t *x, const uint64_t *y, int64_t count, uint64_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(count >= 0, "count must be >= 0");
    if (count == 0) return MHX_OK;
    MHX_REQUIRE(x && y && out, "NULL host pointer");
    if (int rc = ctx->activate()) return rc;
    const size_t bytes = sizeof(uint64_t) * (size_t)count;
    if (int rc = ctx->ensure_scratch(0, bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, bytes)) return rc;
    uint64_t *dx = (uint64_t *)ctx->scratch[0], *dy = (uint64_t *)ctx->scratch[2];
    MHX_HIP_CHECK(hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_minhash_merge(ctx, dx, dy, count, dx)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, dx, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

// ---- packing -------------------------------------------------------------------------------
int mhx_bbit_num_blocks(int32_t num_perm, int32_t b, int32_t *num_blocks) {
    if (!num_blocks) return fail(MHX_ERR_INVALID, "num_blocks is NULL");
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be an integer in [0, 32]");
    MHX_REQUIRE(num_perm > 0, "num_perm must be positive");
    const int per = 64 / mhx::bbit_slot_size(b);
    *num_blocks = (num_perm + per - 1) / per;
    return MHX_OK;
}

int mhx_bbit_pack_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t b,
                      uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be an integer in [0, 32]");
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_bbit_pack(ctx, d_sig, MHX_U64, n, k, b, d_out);
}

int mhx_bbit_pack_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b,
                            uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be an integer in [0, 32]");
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_bbit_pack(ctx, d_sig, sig_dtype, n, k, b, d_out);
}

int mhx_band_digests_layout_dev(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands,
                                int32_t r, int layout, uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    MHX_REQUIRE(layout == MHX_ROW_MAJOR || layout == MHX_BAND_MAJOR, "bad layout %d", layout);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_band_digests(ctx, d_sig, sig_dtype, n, k, bands, r, d_out, layout);
}

int mhx_band_digests_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands,
                               int32_t r, uint64_t *d_out) {
    return mhx_band_digests_layout_dev(ctx, d_sig, sig_dtype, n, k, bands, r, MHX_ROW_MAJOR, d_out);
}

// b-bit blocks and band digests of the same matrix: one read when the shape allows the fused kernel, the two kernels otherwise
int mhx_bbit_pack_band_digests_dev(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t b,
                                   int32_t bands, int32_t r, int digest_layout, uint64_t *d_blocks, uint64_t *d_digests, int *fused) {
    if (fused) *fused = 0;
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(b >= 1 && b <= 32, "b must be in [1, 32]");
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0 && k > 0, "bad shape");
    MHX_REQUIRE(digest_layout == MHX_ROW_MAJOR || digest_layout == MHX_BAND_MAJOR, "bad layout %d", digest_layout);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_blocks && d_digests, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    bool done = false;
    if (ctx->opt_pack_fused != 1)
        if (int rc = mhx::launch_bbit_digest_fused(ctx, d_sig, sig_dtype, n, k, b, bands, r, d_blocks, d_digests, digest_layout, &done)) return rc;
    if (fused) *fused = done ? 1 : 0;
    if (done) return MHX_OK;
    if (int rc = mhx::launch_bbit_pack(ctx, d_sig, sig_dtype, n, k, b, d_blocks)) return rc;
    return mhx::launch_band_digests(ctx, d_sig, sig_dtype, n, k, bands, r, d_digests, digest_layout);
}

int mhx_lsh_sort_bands_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int32_t bands,
                                 int32_t r, uint64_t *d_sorted_digests, uint32_t *d_sorted_rows) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_sorted_digests && d_sorted_rows, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lsh_sort_bands(ctx, d_sig, sig_dtype, n, k, bands, r, d_sorted_digests, d_sorted_rows);
}

int mhx_jaccard_pairs_dev_typed(mhx_ctx *ctx, const void *d_sig_a, const void *d_sig_b, int sig_dtype, int32_t k,
                                const int64_t *d_pairs, int64_t n_pairs, int32_t *d_counts) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(k > 0 && n_pairs >= 0, "bad shape");
    if (n_pairs == 0) return MHX_OK;
    MHX_REQUIRE(d_sig_a && d_sig_b && d_pairs && d_counts, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_jaccard_pairs(ctx, d_sig_a, d_sig_b, sig_dtype, k, d_pairs, n_pairs, d_counts);
}

int mhx_band_keys_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands,
                      int32_t r, uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_band_keys(ctx, d_sig, n, k, bands, r, d_out);
}

int mhx_lean_serialize_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int64_t seed,
                           uint8_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lean_serialize(ctx, d_sig, MHX_U64, n, k, seed, 0, d_out);
}

int mhx_lean_serialize_dev_typed(mhx_ctx *ctx, const void *d_sig, int sig_dtype, int64_t n, int32_t k, int64_t seed, int byteorder,
                                 uint8_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "unknown sig_dtype %d", sig_dtype);
    MHX_REQUIRE(byteorder == MHX_LITTLE_ENDIAN || byteorder == MHX_BIG_ENDIAN, "unknown byte order %d", byteorder);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lean_serialize(ctx, d_sig, sig_dtype, n, k, seed, byteorder, d_out);
}

int mhx_lean_deserialize_dev(mhx_ctx *ctx, const uint8_t *d_records, int64_t n, int32_t k, int byteorder, int sig_dtype, void *d_sig,
                             int64_t *d_seeds, uint32_t *d_bad) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "unknown sig_dtype %d", sig_dtype);
    MHX_REQUIRE(byteorder == MHX_LITTLE_ENDIAN || byteorder == MHX_BIG_ENDIAN, "unknown byte order %d", byteorder);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_records && d_sig, "NULL device pointer");
    MHX_REQUIRE(((uintptr_t)d_records & 3) == 0, "records must be 4-byte aligned");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lean_deserialize(ctx, d_records, n, k, byteorder, sig_dtype, d_sig, d_seeds, d_bad);
}

int mhx_bbit_unpack_dev(mhx_ctx *ctx, const uint64_t *d_blocks, int64_t n, int32_t k, int32_t b, uint32_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be in [0, 32]");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_blocks && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_bbit_unpack(ctx, d_blocks, n, k, b, d_out);
}

// host wrappers: stage signature matrix in scratch[0], result in scratch[2]
static int stage_sig(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, size_t out_bytes) {
    if (int rc = ctx->activate()) return rc;
    const size_t in_bytes = sizeof(uint64_t) * (size_t)(n * k);
    if (int rc = ctx->ensure_scratch(0, in_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[0], sig, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    return MHX_OK;
}

static int fetch_out(mhx_ctx *ctx, void *out, size_t out_bytes) {
    MHX_HIP_CHECK(hipMemcpyAsync(out, ctx->scratch[2], out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_bbit_pack(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int32_t b, uint64_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    int32_t nb = 0;
    if (int rc = mhx_bbit_num_blocks(k, b, &nb)) return rc;
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && out, "NULL host pointer");
    const size_t out_bytes = sizeof(uint64_t) * (size_t)(n * nb);
    if (int rc = stage_sig(ctx, sig, n, k, out_bytes)) return rc;
    if (int rc = mhx::launch_bbit_pack(ctx, ctx->scratch[0], MHX_U64, n, k, b, (uint64_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, out, out_bytes);
}

int mhx_band_keys(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                  uint64_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && out, "NULL host pointer");
    const size_t out_bytes = sizeof(uint64_t) * (size_t)(n * bands * r);
    if (int rc = stage_sig(ctx, sig, n, k, out_bytes)) return rc;
    if (int rc = mhx::launch_band_keys(ctx, (const uint64_t *)ctx->scratch[0], n, k, bands, r,
                                       (uint64_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, out, out_bytes);
}

int mhx_band_digests_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                         uint64_t *d_out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_out, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_band_digests(ctx, d_sig, MHX_U64, n, k, bands, r, d_out);
}

int mhx_band_digests(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                     uint64_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && out, "NULL host pointer");
    const size_t out_bytes = sizeof(uint64_t) * (size_t)(n * bands);
    if (int rc = stage_sig(ctx, sig, n, k, out_bytes)) return rc;
    if (int rc = mhx::launch_band_digests(ctx, ctx->scratch[0], MHX_U64, n, k, bands, r,
                                          (uint64_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, out, out_bytes);
}

int mhx_lsh_sort_digests_layout_dev(mhx_ctx *ctx, const uint64_t *d_digests, int64_t n, int32_t bands, int layout, uint64_t *d_sorted_digests,
                             uint32_t *d_sorted_rows) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n >= 0 && bands > 0, "bad shape");
    MHX_REQUIRE(layout == MHX_ROW_MAJOR || layout == MHX_BAND_MAJOR, "bad layout %d", layout);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_digests && d_sorted_digests && d_sorted_rows, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lsh_sort_bands(ctx, d_digests, layout == MHX_BAND_MAJOR ? mhx::kSigDigestsBM : mhx::kSigDigests, n, bands, bands, 1, d_sorted_digests, d_sorted_rows);
}

int mhx_lsh_sort_digests_dev(mhx_ctx *ctx, const uint64_t *d_digests, int64_t n, int32_t bands, uint64_t *d_sorted_digests,
                             uint32_t *d_sorted_rows) {
    return mhx_lsh_sort_digests_layout_dev(ctx, d_digests, n, bands, MHX_ROW_MAJOR, d_sorted_digests, d_sorted_rows);
}

int mhx_lsh_sort_bands_dev(mhx_ctx *ctx, const uint64_t *d_sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                           uint64_t *d_sorted_digests, uint32_t *d_sorted_rows) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sig && d_sorted_digests && d_sorted_rows, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lsh_sort_bands(ctx, d_sig, MHX_U64, n, k, bands, r, d_sorted_digests, d_sorted_rows);
}

int mhx_lsh_sort_bands(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                       uint64_t *sorted_digests, uint32_t *sorted_rows) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && sorted_digests && sorted_rows, "NULL host pointer");
    const size_t dig_bytes = sizeof(uint64_t) * (size_t)n * bands, row_bytes = sizeof(uint32_t) * (size_t)n * bands;
    const size_t dig_pad = (dig_bytes + 255) & ~(size_t)255;
    if (int rc = stage_sig(ctx, sig, n, k, dig_pad + row_bytes)) return rc;
    uint64_t *d_dig = (uint64_t *)ctx->scratch[2];
    uint32_t *d_rows = (uint32_t *)((char *)ctx->scratch[2] + dig_pad);
    if (int rc = mhx::launch_lsh_sort_bands(ctx, ctx->scratch[0], MHX_U64, n, k, bands, r, d_dig, d_rows)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(sorted_digests, d_dig, dig_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(sorted_rows, d_rows, row_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_lsh_candidate_pairs_dev(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n,
                                int32_t bands, int64_t *d_pairs, int64_t capacity, int64_t *n_pairs, int64_t *n_raw) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_pairs, "n_pairs is NULL");
    MHX_REQUIRE(bands > 0 && n >= 0 && capacity >= 0, "bad shape");
    MHX_REQUIRE(n < ((int64_t)1 << 32), "more than 2^32-1 signatures per call");
    *n_pairs = 0;
    if (n_raw) *n_raw = 0;
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(d_sorted_digests && d_sorted_rows && (d_pairs || capacity == 0), "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lsh_candidate_pairs(ctx, d_sorted_digests, d_sorted_rows, n, bands, d_pairs, capacity, n_pairs, n_raw);
}

int mhx_lsh_candidate_pairs(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int32_t bands, int32_t r,
                            int64_t *pairs, int64_t capacity, int64_t *n_pairs, int64_t *n_raw) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_pairs, "n_pairs is NULL");
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0 && capacity >= 0, "bad shape");
    *n_pairs = 0;
    if (n_raw) *n_raw = 0;
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && (pairs || capacity == 0), "NULL host pointer");
    // scratch[2]: sorted digests | sorted rows | pairs
    const size_t dig_bytes = ((sizeof(uint64_t) * (size_t)n * bands) + 255) & ~(size_t)255;
    const size_t row_bytes = ((sizeof(uint32_t) * (size_t)n * bands) + 255) & ~(size_t)255;
    const size_t pair_bytes = sizeof(int64_t) * 2 * (size_t)capacity;
    if (int rc = stage_sig(ctx, sig, n, k, dig_bytes + row_bytes + pair_bytes)) return rc;
    uint64_t *d_dig = (uint64_t *)ctx->scratch[2];
    uint32_t *d_rows = (uint32_t *)((char *)ctx->scratch[2] + dig_bytes);
    int64_t *d_pairs = (int64_t *)((char *)ctx->scratch[2] + dig_bytes + row_bytes);
    if (int rc = mhx::launch_lsh_sort_bands(ctx, ctx->scratch[0], MHX_U64, n, k, bands, r, d_dig, d_rows)) return rc;
    if (int rc = mhx::launch_lsh_candidate_pairs(ctx, d_dig, d_rows, n, bands, d_pairs, capacity, n_pairs, n_raw)) return rc;
    if (*n_pairs > 0 && *n_pairs <= capacity)
        MHX_HIP_CHECK(hipMemcpyAsync(pairs, d_pairs, sizeof(int64_t) * 2 * (size_t)*n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_lsh_query_dev(mhx_ctx *ctx, const uint64_t *d_sorted_digests, const uint32_t *d_sorted_rows, int64_t n,
                      int32_t bands, int32_t r, const void *d_query_sig, const void *d_index_sig, int sig_dtype,
                      int32_t k, int64_t m, int64_t *d_pairs, int64_t capacity, int64_t *n_pairs) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_pairs, "n_pairs is NULL");
    MHX_REQUIRE(sig_dtype == MHX_U64 || sig_dtype == MHX_U32, "bad sig_dtype %d", sig_dtype);
    MHX_REQUIRE(bands > 0 && r > 0 && (int64_t)bands * r <= k, "bands*r must be in (0, num_perm]");
    MHX_REQUIRE(n >= 0 && m >= 0 && capacity >= 0, "bad shape");
    MHX_REQUIRE(n < ((int64_t)1 << 32) && m < ((int64_t)1 << 32), "more than 2^32-1 rows per call");
    *n_pairs = 0;
    if (n == 0 || m == 0) return MHX_OK;
    MHX_REQUIRE(d_sorted_digests && d_sorted_rows && d_query_sig && (d_pairs || capacity == 0), "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_lsh_query(ctx, d_sorted_digests, d_sorted_rows, n, bands, r, d_query_sig, d_index_sig, sig_dtype, k, m,
                                 d_pairs, capacity, n_pairs);
}

int mhx_jaccard_pairs_dev(mhx_ctx *ctx, const uint64_t *d_sig_a, const uint64_t *d_sig_b, int32_t k,
                          const int64_t *d_pairs, int64_t n_pairs, int32_t *d_counts) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n_pairs >= 0, "bad shape");
    if (n_pairs == 0) return MHX_OK;
    MHX_REQUIRE(d_sig_a && d_sig_b && d_pairs && d_counts, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_jaccard_pairs(ctx, d_sig_a, d_sig_b, MHX_U64, k, d_pairs, n_pairs, d_counts);
}

int mhx_jaccard_pairs(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, const int64_t *pairs,
                      int64_t n_pairs, int32_t *counts) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0 && n_pairs >= 0, "bad shape");
    if (n_pairs == 0) return MHX_OK;
    MHX_REQUIRE(sig && pairs && counts, "NULL host pointer");
    for (int64_t p = 0; p < 2 * n_pairs; ++p)
        MHX_REQUIRE(pairs[p] >= 0 && pairs[p] < n, "pair index %lld out of range [0,%lld)", (long long)pairs[p], (long long)n);
    const size_t out_bytes = sizeof(int32_t) * (size_t)n_pairs;
    if (int rc = stage_sig(ctx, sig, n, k, out_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(1, sizeof(int64_t) * 2 * (size_t)n_pairs)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[1], pairs, sizeof(int64_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice,
                                 ctx->stream));
    const uint64_t *d_sig = (const uint64_t *)ctx->scratch[0];
    if (int rc = mhx::launch_jaccard_pairs(ctx, d_sig, d_sig, MHX_U64, k, (const int64_t *)ctx->scratch[1], n_pairs,
                                           (int32_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, counts, out_bytes);
}

int mhx_bbit_jaccard_pairs_dev(mhx_ctx *ctx, const uint64_t *d_blocks_a, const uint64_t *d_blocks_b, int32_t k, int32_t b,
                               const int64_t *d_pairs, int64_t n_pairs, int32_t *d_counts) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be an integer in [0, 32]");
    MHX_REQUIRE(k > 0 && n_pairs >= 0, "bad shape");
    if (n_pairs == 0) return MHX_OK;
    MHX_REQUIRE(d_blocks_a && d_blocks_b && d_pairs && d_counts, "NULL device pointer");
    if (int rc = ctx->activate()) return rc;
    return mhx::launch_bbit_jaccard(ctx, d_blocks_a, d_blocks_b, k, b, d_pairs, n_pairs, d_counts);
}

int mhx_bbit_jaccard_pairs(mhx_ctx *ctx, const uint64_t *blocks, int64_t n, int32_t k, int32_t b, const int64_t *pairs,
                           int64_t n_pairs, int32_t *counts) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    int32_t nb = 0;
    if (int rc = mhx_bbit_num_blocks(k, b, &nb)) return rc;
    MHX_REQUIRE(n >= 0 && n_pairs >= 0, "bad shape");
    if (n_pairs == 0) return MHX_OK;
    MHX_REQUIRE(blocks && pairs && counts, "NULL host pointer");
    for (int64_t p = 0; p < 2 * n_pairs; ++p)
        MHX_REQUIRE(pairs[p] >= 0 && pairs[p] < n, "pair index %lld out of range [0,%lld)", (long long)pairs[p], (long long)n);
    if (int rc = ctx->activate()) return rc;
    const size_t in_bytes = sizeof(uint64_t) * (size_t)n * (size_t)nb, pair_bytes = sizeof(int64_t) * 2 * (size_t)n_pairs;
    const size_t out_bytes = sizeof(int32_t) * (size_t)n_pairs;
    if (int rc = ctx->ensure_scratch(0, in_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(1, pair_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[0], blocks, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[1], pairs, pair_bytes, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t *d_blocks = (const uint64_t *)ctx->scratch[0];
    if (int rc = mhx::launch_bbit_jaccard(ctx, d_blocks, d_blocks, k, b, (const int64_t *)ctx->scratch[1], n_pairs,
                                          (int32_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, counts, out_bytes);
}

int mhx_lean_serialize(mhx_ctx *ctx, const uint64_t *sig, int64_t n, int32_t k, int64_t seed, uint8_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(sig && out, "NULL host pointer");
    const size_t out_bytes = (size_t)n * (12 + 4 * (size_t)k);
    if (int rc = stage_sig(ctx, sig, n, k, out_bytes)) return rc;
    if (int rc = mhx::launch_lean_serialize(ctx, ctx->scratch[0], MHX_U64, n, k, seed, 0, (uint8_t *)ctx->scratch[2]))
        return rc;
    return fetch_out(ctx, out, out_bytes);
}

// records (host) -> [n, k] uint64 hashvalues + seeds; a record whose length field is not k: MHX_ERR_INVALID, nothing written
int mhx_lean_deserialize(mhx_ctx *ctx, const uint8_t *records, int64_t n, int32_t k, int byteorder, uint64_t *sig, int64_t *seeds) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    MHX_REQUIRE(byteorder == MHX_LITTLE_ENDIAN || byteorder == MHX_BIG_ENDIAN, "unknown byte order %d", byteorder);
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(records && sig, "NULL host pointer");
    if (int rc = ctx->activate()) return rc;
    const size_t rec_bytes = (size_t)n * (12 + 4 * (size_t)k), sig_bytes = sizeof(uint64_t) * (size_t)n * k;
    const size_t seeds_at = (sig_bytes + 255) & ~(size_t)255, bad_at = seeds_at + (((size_t)n * 8 + 255) & ~(size_t)255);
    if (int rc = ctx->ensure_scratch(0, rec_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, bad_at + 256)) return rc;
    char *base = (char *)ctx->scratch[2];
    unsigned int *d_bad = (unsigned int *)(base + bad_at);
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[0], records, rec_bytes, hipMemcpyHostToDevice, ctx->stream));
    MHX_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(unsigned int), ctx->stream));
    if (int rc = mhx::launch_lean_deserialize(ctx, (const uint8_t *)ctx->scratch[0], n, k, byteorder, MHX_U64, base, (int64_t *)(base + seeds_at), d_bad))
        return rc;
    unsigned int bad = 0;
    MHX_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (bad) return fail(MHX_ERR_INVALID, "%u of %lld records do not hold %d hash values (length field)", bad, (long long)n, k);
    MHX_HIP_CHECK(hipMemcpyAsync(sig, base, sig_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (seeds) MHX_HIP_CHECK(hipMemcpyAsync(seeds, base + seeds_at, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_bbit_unpack(mhx_ctx *ctx, const uint64_t *blocks, int64_t n, int32_t k, int32_t b, uint32_t *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(k > 0 && n >= 0, "bad shape");
    MHX_REQUIRE(b >= 0 && b <= 32, "b must be in [0, 32]");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(blocks && out, "NULL host pointer");
    if (int rc = ctx->activate()) return rc;
    int32_t nb = 0;
    if (int rc = mhx_bbit_num_blocks(k, b, &nb)) return rc;
    const size_t in_bytes = sizeof(uint64_t) * (size_t)n * nb, out_bytes = sizeof(uint32_t) * (size_t)n * k;
    if (int rc = ctx->ensure_scratch(0, in_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, out_bytes)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[0], blocks, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_bbit_unpack(ctx, (const uint64_t *)ctx->scratch[0], n, k, b, (uint32_t *)ctx->scratch[2])) return rc;
    return fetch_out(ctx, out, out_bytes);
}

// ---- weighted ------------------------------------------------------------------------------
int mhx_wgen_create(mhx_ctx *ctx, const float *rs, const float *ln_cs, const float *betas,
                    int32_t sample_size, int32_t dim, mhx_wgen **out) {
    if (!ctx || !rs || !ln_cs || !betas || !out) return fail(MHX_ERR_INVALID, "NULL argument");
    MHX_GUARD(ctx);
    MHX_REQUIRE(sample_size > 0 && dim > 0, "sample_size and dim must be positive");
    if (int rc = ctx->activate()) return rc;
    mhx_wgen *g = new mhx_wgen();
    g->ctx = ctx;
    g->sample_size = sample_size;
    g->dim = dim;
    g->s_pad = (sample_size + 63) / 64 * 64;
    const size_t n = (size_t)sample_size * dim;
    const size_t t_bytes = sizeof(float) * 5 * (size_t)g->s_pad * dim;
    g->table_fast = true;
    for (size_t j = 0; j < n && g->table_fast; ++j) {
        const float m = fabsf(rs[j]);
        g->table_fast = m >= 0x1p-40f && m <= 0x1p40f;  // false for NaN / inf / 0 as well
    }
    // the walk's bound wants r > 0 and finite ln_c, beta (monotone ln_a); its table builder sorts a sample's columns in LDS
    g->walk_ok = g->table_fast && dim <= 16384;
    for (size_t j = 0; j < n && g->walk_ok; ++j)
        g->walk_ok = rs[j] > 0.0f && fabsf(ln_cs[j]) < __builtin_inff() && fabsf(betas[j]) < __builtin_inff();
    const size_t a_bytes = sizeof(float) * 4 * (size_t)g->s_pad * (size_t)dim;
    const float plan0[8] = {__builtin_nanf(""), 0.0f, 0.0f, 0.0f, __builtin_nanf(""), 0.0f, 0.0f, 0.0f};  // two WalkPlan records: no tables yet
    hipError_t e = mhx::dev_malloc((void **)&g->d_params, t_bytes);
    if (e == hipSuccess) e = mhx::dev_malloc((void **)&g->d_aos, a_bytes);
    if (e == hipSuccess && g->walk_ok) e = mhx::dev_malloc((void **)&g->d_walk_a, a_bytes);
    if (e == hipSuccess && g->walk_ok) e = mhx::dev_malloc((void **)&g->d_walk_c, a_bytes / 4);
    if (e == hipSuccess && g->walk_ok) e = mhx::dev_malloc(&g->d_walk_plan, sizeof(plan0));
    if (e == hipSuccess && g->walk_ok) e = hipMemcpyAsync(g->d_walk_plan, plan0, sizeof(plan0), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && g->walk_ok) e = hipMemsetAsync(g->d_walk_a, 0, a_bytes, ctx->stream);  // lanes behind sample_size load from here too
    if (e == hipSuccess && g->walk_ok) e = hipMemsetAsync(g->d_walk_c, 0, a_bytes / 4, ctx->stream);
    if (e != hipSuccess) {
        (void)mhx::dev_free(g->d_params);
        (void)mhx::dev_free(g->d_aos);
        (void)mhx::dev_free(g->d_walk_a);
        (void)mhx::dev_free(g->d_walk_c);
        (void)mhx::dev_free(g->d_walk_plan);
        delete g;
        return fail(MHX_ERR_OOM, "hipMalloc for weighted parameters failed: %s", hipGetErrorString(e));
    }
    int rc = ctx->ensure_scratch(0, 3 * n * sizeof(float));
    if (rc == MHX_OK) {
        float *d = (float *)ctx->scratch[0];
        e = hipMemcpyAsync(d, rs, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + n, ln_cs, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d + 2 * n, betas, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) rc = fail(MHX_ERR_HIP, "uploading weighted parameters failed: %s", hipGetErrorString(e));
        if (rc == MHX_OK) rc = mhx::launch_wgen_transpose(g, d, d + n, d + 2 * n);
        if (rc == MHX_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(MHX_ERR_HIP, "weighted parameter transpose failed");
    }
    if (rc != MHX_OK) {
        (void)mhx::dev_free(g->d_params);
        (void)mhx::dev_free(g->d_aos);
        (void)mhx::dev_free(g->d_walk_a);
        (void)mhx::dev_free(g->d_walk_c);
        (void)mhx::dev_free(g->d_walk_plan);
        delete g;
        return rc;
    }
    *out = g;
    return MHX_OK;
}

int mhx_wgen_destroy(mhx_wgen *gen) {
    if (!gen) return MHX_OK;
    MHX_GUARD(gen->ctx);
    (void)hipSetDevice(gen->ctx->device);
    (void)hipStreamSynchronize(gen->ctx->stream);
    (void)mhx::dev_free(gen->d_params);
    (void)mhx::dev_free(gen->d_aos);
    (void)mhx::dev_free(gen->d_walk_a);
    (void)mhx::dev_free(gen->d_walk_c);
    (void)mhx::dev_free(gen->d_walk_plan);
    delete gen;
    return MHX_OK;
}

int mhx_weighted_minhash_many_dev(mhx_wgen *gen, const int64_t *d_indptr, const int32_t *d_indices,
                                  const float *d_values, int values_are_logs, int64_t n_rows,
                                  int64_t nnz, int64_t *d_out, uint8_t *d_nonempty) {
    if (!gen) return fail(MHX_ERR_INVALID, "gen is NULL");
    MHX_GUARD(gen->ctx);
    MHX_REQUIRE(n_rows >= 0 && nnz >= 0, "bad shape");
    if (n_rows == 0) return MHX_OK;
    MHX_REQUIRE(d_indptr && d_out && d_nonempty, "NULL device pointer");
    MHX_REQUIRE((d_indices && d_values) || nnz == 0, "NULL device pointer");
    if (int rc = gen->ctx->activate()) return rc;
    return mhx::launch_weighted(gen, d_indptr, d_indices, d_values, values_are_logs, n_rows, nnz, d_out,
                                d_nonempty);
}

int mhx_weighted_minhash_many(mhx_wgen *gen, const int64_t *indptr, const int32_t *indices,
                              const float *values, int values_are_logs, int64_t n_rows, int64_t *out,
                              uint8_t *nonempty) {
    if (!gen) return fail(MHX_ERR_INVALID, "gen is NULL");
    MHX_GUARD(gen->ctx);
    MHX_REQUIRE(n_rows >= 0, "bad shape");
    if (n_rows == 0) return MHX_OK;
    MHX_REQUIRE(indptr && out && nonempty, "NULL host pointer");
    mhx_ctx *ctx = gen->ctx;
    if (int rc = ctx->activate()) return rc;
    for (int64_t i = 0; i < n_rows; ++i)
        MHX_REQUIRE(indptr[i + 1] >= indptr[i], "indptr must be non-decreasing (row %lld)", (long long)i);
    MHX_REQUIRE(indptr[0] == 0, "indptr[0] must be 0");
    const int64_t nnz = indptr[n_rows];
    MHX_REQUIRE((indices && values) || nnz == 0, "NULL host pointer");
    for (int64_t j = 0; j < nnz; ++j)
        MHX_REQUIRE(indices[j] >= 0 && indices[j] < gen->dim, "column index %d out of range [0,%d)", indices[j], gen->dim);
    const size_t ptr_bytes = sizeof(int64_t) * (size_t)(n_rows + 1);
    const size_t idx_bytes = ((sizeof(int32_t) * (size_t)nnz) + 255) & ~(size_t)255;
    const size_t val_bytes = sizeof(float) * (size_t)nnz;
    const size_t out_bytes = sizeof(int64_t) * 2 * (size_t)gen->sample_size * (size_t)n_rows;
    const size_t ne_off = (out_bytes + 255) & ~(size_t)255;
    if (int rc = ctx->ensure_scratch(0, idx_bytes + val_bytes + 256)) return rc;
    if (int rc = ctx->ensure_scratch(1, ptr_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, ne_off + (size_t)n_rows)) return rc;
    int32_t *d_idx = (int32_t *)ctx->scratch[0];
    float *d_val = (float *)((char *)ctx->scratch[0] + idx_bytes);
    int64_t *d_ptr = (int64_t *)ctx->scratch[1];
    int64_t *d_out = (int64_t *)ctx->scratch[2];
    uint8_t *d_ne = (uint8_t *)ctx->scratch[2] + ne_off;
    MHX_HIP_CHECK(hipMemcpyAsync(d_ptr, indptr, ptr_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (nnz) {
        MHX_HIP_CHECK(hipMemcpyAsync(d_idx, indices, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
        MHX_HIP_CHECK(hipMemcpyAsync(d_val, values, val_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if (int rc = mhx::launch_weighted(gen, d_ptr, d_idx, d_val, values_are_logs, n_rows, nnz, d_out, d_ne))
        return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(nonempty, d_ne, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_weighted_logf(mhx_ctx *ctx, const float *x, int64_t n, float *out) {
    if (!ctx) return fail(MHX_ERR_INVALID, "ctx is NULL");
    MHX_GUARD(ctx);
    MHX_REQUIRE(n >= 0, "bad shape");
    if (n == 0) return MHX_OK;
    MHX_REQUIRE(x && out, "NULL host pointer");
    if (int rc = ctx->activate()) return rc;
    const size_t bytes = sizeof(float) * (size_t)n;
    if (int rc = ctx->ensure_scratch(0, bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, bytes)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(ctx->scratch[0], x, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_weighted_log(ctx, (const float *)ctx->scratch[0], n, (float *)ctx->scratch[2])) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, ctx->scratch[2], bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

int mhx_weighted_minhash_many_dense_dev(mhx_wgen *gen, const float *d_x, int values_are_logs, int64_t n_rows,
                                        int64_t *d_out, uint8_t *d_nonempty) {
    if (!gen) return fail(MHX_ERR_INVALID, "gen is NULL");
    MHX_GUARD(gen->ctx);
    MHX_REQUIRE(n_rows >= 0, "bad shape");
    if (n_rows == 0) return MHX_OK;
    MHX_REQUIRE(d_x && d_out && d_nonempty, "NULL device pointer");
    if (int rc = gen->ctx->activate()) return rc;
    return mhx::launch_weighted_dense(gen, d_x, values_are_logs, n_rows, d_out, d_nonempty);
}

}  // extern "C" (the feed's state and helpers are C++)

// A dense weighted call in pieces: two device slots, so that the upload of piece i+1 (copy_in), the evaluation of
// piece i (ctx->stream) and the download of piece i-1 (copy_out) run side by side.
struct mhx_wfeed {
    mhx_wgen *gen = nullptr;
    int values_are_logs = 0;
    int64_t piece_rows = 0;
    float *d_x[2] = {nullptr, nullptr};
    char *d_res[2] = {nullptr, nullptr};  // out int64[piece_rows, S, 2] | nonempty uint8[piece_rows]
    size_t ne_off = 0;
    hipEvent_t uploaded[2] = {nullptr, nullptr}, computed[2] = {nullptr, nullptr};
    int64_t fed = 0;
    // the piece fed last: evaluated (or being evaluated) on the device, results not yet on the host
    int64_t *pend_out = nullptr;
    uint8_t *pend_ne = nullptr;
    int64_t pend_rows = 0;
    int pend_slot = 0;
};

namespace {

// bring the pending piece down (blocks until it is on the host)
int feed_drain(mhx_wfeed *f) {
    if (!f->pend_rows) return MHX_OK;
    mhx_ctx *ctx = f->gen->ctx;
    const int slot = f->pend_slot;
    const size_t out_bytes = sizeof(int64_t) * 2 * (size_t)f->gen->sample_size * (size_t)f->pend_rows;
    const int64_t rows = f->pend_rows;
    f->pend_rows = 0;
    MHX_HIP_CHECK(hipStreamWaitEvent(ctx->copy_out, f->computed[slot], 0));
    MHX_HIP_CHECK(hipMemcpyAsync(f->pend_out, f->d_res[slot], out_bytes, hipMemcpyDeviceToHost, ctx->copy_out));
    MHX_HIP_CHECK(hipMemcpyAsync(f->pend_ne, f->d_res[slot] + f->ne_off, (size_t)rows, hipMemcpyDeviceToHost, ctx->copy_out));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->copy_out));
    return MHX_OK;
}

// rows per piece of a one-call dense evaluation: about 64 MiB of input + output, at least 4096 rows (fewer leave
// CUs without a row block), whole row blocks of 8
int64_t dense_piece_rows(const mhx_wgen *gen) {
    const int64_t per_row = 4 * (int64_t)gen->dim + 16 * (int64_t)gen->sample_size;
    return std::max<int64_t>(4096, ((64ll << 20) / per_row) & ~7ll);
}

}  // namespace

extern "C" {

int mhx_weighted_dense_begin(mhx_wgen *gen, int values_are_logs, int64_t piece_rows, mhx_wfeed **feed) {
    if (!gen) return fail(MHX_ERR_INVALID, "gen is NULL");
    MHX_GUARD(gen->ctx);
    MHX_REQUIRE(feed, "feed is NULL");
    *feed = nullptr;
    MHX_REQUIRE(piece_rows > 0, "piece_rows must be positive");
    mhx_ctx *ctx = gen->ctx;
    if (int rc = ctx->activate()) return rc;
    if (int rc = ctx->ensure_copy_streams()) return rc;
    mhx_wfeed *f = new (std::nothrow) mhx_wfeed();
    if (!f) return fail(MHX_ERR_OOM, "out of host memory");
    f->gen = gen;
    f->values_are_logs = values_are_logs;
    f->piece_rows = piece_rows;
    const size_t x_bytes = sizeof(float) * (size_t)piece_rows * (size_t)gen->dim;
    const size_t out_bytes = sizeof(int64_t) * 2 * (size_t)gen->sample_size * (size_t)piece_rows;
    f->ne_off = (out_bytes + 255) & ~(size_t)255;
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = mhx::dev_malloc(reinterpret_cast<void **>(&f->d_x[i]), x_bytes);
        if (e == hipSuccess) e = mhx::dev_malloc(reinterpret_cast<void **>(&f->d_res[i]), f->ne_off + (size_t)piece_rows);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->uploaded[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&f->computed[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        (void)mhx_weighted_dense_end(f);
        return fail(e == hipErrorOutOfMemory ? MHX_ERR_OOM : MHX_ERR_HIP, "buffers for pieces of %lld rows: %s", (long long)piece_rows,
                    hipGetErrorString(e));
    }
    *feed = f;
    return MHX_OK;
}

int mhx_weighted_dense_feed(mhx_wfeed *f, const float *x, int64_t n_rows, int64_t *out, uint8_t *nonempty) {
    if (!f) return fail(MHX_ERR_INVALID, "feed is NULL");
    mhx_ctx *ctx = f->gen->ctx;
    MHX_GUARD(ctx);
    MHX_REQUIRE(n_rows >= 0 && n_rows <= f->piece_rows, "a piece holds at most %lld rows", (long long)f->piece_rows);
    if (n_rows == 0) return MHX_OK;
    MHX_REQUIRE(x && out && nonempty, "NULL host pointer");
    if (int rc = ctx->activate()) return rc;
    const int slot = (int)(f->fed & 1);
    // the slot's input was read by the evaluation of the piece before last; its results came down during the last feed
    if (f->fed >= 2) MHX_HIP_CHECK(hipEventSynchronize(f->computed[slot]));
    const size_t x_bytes = sizeof(float) * (size_t)n_rows * (size_t)f->gen->dim;
    MHX_HIP_CHECK(hipMemcpyAsync(f->d_x[slot], x, x_bytes, hipMemcpyHostToDevice, ctx->copy_in));
    MHX_HIP_CHECK(hipEventRecord(f->uploaded[slot], ctx->copy_in));
    MHX_HIP_CHECK(hipStreamWaitEvent(ctx->stream, f->uploaded[slot], 0));
    if (int rc = mhx::launch_weighted_dense(f->gen, f->d_x[slot], f->values_are_logs, n_rows, reinterpret_cast<int64_t *>(f->d_res[slot]),
                                            reinterpret_cast<uint8_t *>(f->d_res[slot] + f->ne_off)))
        return rc;
    MHX_HIP_CHECK(hipEventRecord(f->computed[slot], ctx->stream));
    ++f->fed;
    if (int rc = feed_drain(f)) return rc;  // the previous piece, while this one is evaluated
    f->pend_out = out;
    f->pend_ne = nonempty;
    f->pend_rows = n_rows;
    f->pend_slot = slot;
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->copy_in));  // a pinned x is copied asynchronously: it is free from here on
    return MHX_OK;
}

int mhx_weighted_dense_end(mhx_wfeed *f) {
    if (!f) return MHX_OK;
    mhx_ctx *ctx = f->gen->ctx;
    MHX_GUARD(ctx);
    int rc = ctx->activate();
    if (!rc) rc = feed_drain(f);
    if (ctx->copy_in) (void)hipStreamSynchronize(ctx->copy_in);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i) {
        if (f->d_x[i]) (void)mhx::dev_free(f->d_x[i]);
        if (f->d_res[i]) (void)mhx::dev_free(f->d_res[i]);
        if (f->uploaded[i]) (void)hipEventDestroy(f->uploaded[i]);
        if (f->computed[i]) (void)hipEventDestroy(f->computed[i]);
    }
    delete f;
    return rc;
}

int mhx_weighted_minhash_many_dense(mhx_wgen *gen, const float *x, int values_are_logs, int64_t n_rows, int64_t *out,
                                    uint8_t *nonempty) {
    if (!gen) return fail(MHX_ERR_INVALID, "gen is NULL");
    MHX_GUARD(gen->ctx);
    MHX_REQUIRE(n_rows >= 0, "bad shape");
    if (n_rows == 0) return MHX_OK;
    MHX_REQUIRE(x && out && nonempty, "NULL host pointer");
    mhx_ctx *ctx = gen->ctx;
    if (int rc = ctx->activate()) return rc;
    const int64_t piece = dense_piece_rows(gen);
    if (n_rows >= 2 * piece && ctx->opt_host_chunk_bytes >= 0) {  // upload, evaluation and download side by side
        mhx_wfeed *f = nullptr;
        if (int rc = mhx_weighted_dense_begin(gen, values_are_logs, piece, &f)) return rc;
        int rc = MHX_OK;
        for (int64_t lo = 0; lo < n_rows && !rc; lo += piece) {
            const int64_t rows = std::min(piece, n_rows - lo);
            rc = mhx_weighted_dense_feed(f, x + (size_t)lo * (size_t)gen->dim, rows, out + (size_t)lo * 2 * (size_t)gen->sample_size,
                                         nonempty + lo);
        }
        const int rc_end = mhx_weighted_dense_end(f);
        return rc ? rc : rc_end;
    }
    const size_t x_bytes = sizeof(float) * (size_t)n_rows * (size_t)gen->dim;
    const size_t out_bytes = sizeof(int64_t) * 2 * (size_t)gen->sample_size * (size_t)n_rows;
    const size_t ne_off = (out_bytes + 255) & ~(size_t)255;
    if (int rc = ctx->ensure_scratch(0, x_bytes)) return rc;
    if (int rc = ctx->ensure_scratch(2, ne_off + (size_t)n_rows)) return rc;
    float *d_x = (float *)ctx->scratch[0];
    int64_t *d_out = (int64_t *)ctx->scratch[2];
    uint8_t *d_ne = (uint8_t *)ctx->scratch[2] + ne_off;
    MHX_HIP_CHECK(hipMemcpyAsync(d_x, x, x_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = mhx::launch_weighted_dense(gen, d_x, values_are_logs, n_rows, d_out, d_ne)) return rc;
    MHX_HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipMemcpyAsync(nonempty, d_ne, (size_t)n_rows, hipMemcpyDeviceToHost, ctx->stream));
    MHX_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MHX_OK;
}

}  // extern "C"

// hk_api.hip -- C-ABI host layer of libhomonim_hk.so (declared in include/homonim_hk.h).
//
// Owns: one HIP device per context, a pool of streams each with a device staging slab, argument validation
// (mirrors homonim/utils.py:104-133 and homonim/kernel_model.py:430-431,459-460 error behaviour as status codes),
// and the dispatch to the gfx950 kernels in hk_fit_kernel.h (through hk_kernels.hip) / hk_norm.hip.  No global mutable state besides the
// thread-local error string; any number of host threads may use one context (homonim/fuse.py:396-401).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cctype>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/homonim_hk.h"
#include "../../include/homonim_hk_devtools.h"
#include "hk_kernels.h"

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>  // types only: the loaded HSA runtime is looked up at run time (gpu_fault_report)
#include <rccl/rccl.h>  // types and prototypes only: librccl is opened at run time (hk_comm_*), not linked

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// HIP keeps a per-thread "last error" until somebody reads it, and the launch wrappers of the kernel files report their
// launches through hipGetLastError(): a failure this layer hands back to a caller who carries on (e.g. hk_host_register of
// memory that is page-locked already) would otherwise surface again as the status of that thread's NEXT launch.  So a
// failing call clears the state it leaves behind, and every entry point that selects the device starts from a clean slate.
#define HK_HIP(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            (void)hipGetLastError();                                                                       \
            return fail(HK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
        }                                                                                                  \
    } while (0)
#define HK_ENTER(ctx)                         \
    do {                                      \
        HK_HIP(hipSetDevice((ctx)->device));  \
        (void)hipGetLastError();              \
    } while (0)

// Every device allocation of the library goes through dev_malloc / dev_free.  HK_GUARD_ALLOC=lo|hi (a debugging switch: GPU
// AddressSanitizer is not available everywhere) places each allocation in its own mapping of the virtual-memory API with
// UNMAPPED address ranges on both sides, its first byte at the start of the mapping (lo: reads or writes below the buffer
// fault) or its last 256-byte unit at the end (hi: those above it do), so an out-of-range access of a kernel ends the
// process at that launch instead of reading whatever happens to be mapped next to the buffer.
struct GuardedRange {
    void* user;
    void* va;
    size_t va_bytes;
    void* mapped;
    size_t mapped_bytes;
    hipMemGenericAllocationHandle_t handle;
};
int guard_mode() {  // 0 off, 1 lo, 2 hi
    static const int mode = [] {
        const char* e = getenv("HK_GUARD_ALLOC");
        if (!e || !*e || !strcmp(e, "0")) return 0;
        if (!strcmp(e, "poison")) return 3;  // plain hipMalloc, contents set to 0xAB: nothing may rely on fresh memory being zero
        return !strcmp(e, "hi") ? 2 : 1;
    }();
    return mode;
}
std::mutex g_guard_mu;
std::vector<GuardedRange> g_guarded;

// What the library has allocated on the device (a handful of slabs, workspaces and the callers' hk_dev_alloc buffers): a GPU
// memory fault is reported with its place relative to them (gpu_fault_report).
struct DevRange {
    uintptr_t base;
    size_t bytes;
};
std::mutex g_ranges_mu;
std::vector<DevRange> g_ranges;
void note_alloc(void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    g_ranges.push_back({reinterpret_cast<uintptr_t>(p), bytes});
}
void note_free(void* p) {
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    for (size_t i = 0; i < g_ranges.size(); ++i)
        if (g_ranges[i].base == reinterpret_cast<uintptr_t>(p)) {
            g_ranges[i] = g_ranges.back();
            g_ranges.pop_back();
            return;
        }
}

// A GPU memory fault (or hardware exception) ends the process with an abort() of the HSA runtime's event thread; the runtime names
// the address, not what it belongs to.  The library registers a system-event callback of its own that says where the address
// lies relative to the library's allocations, and then lets the runtime carry on as before (it does not claim to have handled
// the event).
hsa_status_t gpu_fault_report(const hsa_amd_event_t* ev, void*) {
    if (!ev) return HSA_STATUS_ERROR;
    if (ev->event_type == HSA_AMD_GPU_MEMORY_FAULT_EVENT || ev->event_type == HSA_AMD_GPU_MEMORY_ERROR_EVENT) {
        const bool fault = ev->event_type == HSA_AMD_GPU_MEMORY_FAULT_EVENT;
        const uint64_t addr = fault ? ev->memory_fault.virtual_address : ev->memory_error.virtual_address;
        const uint32_t mask = fault ? ev->memory_fault.fault_reason_mask : ev->memory_error.error_reason_mask;
        char where[200] = "not near any allocation of libhomonim_hk";
        {
            std::lock_guard<std::mutex> lk(g_ranges_mu);
            uint64_t best = ~0ull;
            for (const DevRange& r : g_ranges) {
                if (addr >= r.base && addr < r.base + r.bytes) {
                    snprintf(where, sizeof where, "INSIDE a library allocation of %zu bytes (offset %llu): freed or remapped under a kernel?",
                             r.bytes, (unsigned long long)(addr - r.base));
                    best = 0;
                    break;
                }
                const uint64_t d = addr < r.base ? r.base - addr : addr - (r.base + r.bytes) + 1;
                if (d < best && d < (64ull << 20)) {
                    best = d;
                    if (addr < r.base)
                        snprintf(where, sizeof where, "%llu bytes BELOW a library allocation of %zu bytes", (unsigned long long)d, r.bytes);
                    else
                        snprintf(where, sizeof where, "%llu bytes past the END of a library allocation of %zu bytes",
                                 (unsigned long long)(d - 1), r.bytes);
                }
            }
        }
        fprintf(stderr, "[libhomonim_hk] GPU memory %s at 0x%llx, reason mask 0x%x%s%s%s%s: %s\n", fault ? "access fault" : "error",
                (unsigned long long)addr, mask, (fault && (mask & HSA_AMD_MEMORY_FAULT_PAGE_NOT_PRESENT)) ? " page-not-present" : "",
                (fault && (mask & HSA_AMD_MEMORY_FAULT_READ_ONLY)) ? " read-only" : "",
                (fault && (mask & (HSA_AMD_MEMORY_FAULT_DRAMECC | HSA_AMD_MEMORY_FAULT_SRAMECC))) ? " ECC" : "",
                (fault && (mask & HSA_AMD_MEMORY_FAULT_HANG)) ? " hang/reset" : "", where);
        fflush(stderr);
    } else if (ev->event_type == HSA_AMD_GPU_HW_EXCEPTION_EVENT) {
        fprintf(stderr, "[libhomonim_hk] GPU hardware exception: reset type 0x%x, cause 0x%x%s%s\n", (unsigned)ev->hw_exception.reset_type,
                (unsigned)ev->hw_exception.reset_cause,
                (ev->hw_exception.reset_cause & HSA_AMD_HW_EXCEPTION_CAUSE_GPU_HANG) ? " GPU hang" : "",
                (ev->hw_exception.reset_cause & HSA_AMD_HW_EXCEPTION_CAUSE_ECC) ? " ECC" : "");
        fflush(stderr);
    }
    return HSA_STATUS_ERROR;  // not handled here: the runtime's own handling (ending the process) follows
}
void register_gpu_fault_report() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* on = getenv("HK_FAULT_REPORT");  // opt-in: a process-wide callback is not a library call's business
        if (!on || strcmp(on, "1")) return;
        void* hsa = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);  // the instance HIP runs on, or nothing
        if (!hsa) return;
        using reg_t = hsa_status_t (*)(hsa_amd_system_event_callback_t, void*);
        reg_t reg = reinterpret_cast<reg_t>(dlsym(hsa, "hsa_amd_register_system_event_handler"));
        if (reg) (void)reg(gpu_fault_report, nullptr);
    });
}

// does a buffer of `have` bytes serve a request for `need`?  (guarded: only an exact fit, so that the request's end is the mapping's)
bool fits(size_t have, size_t need) { return guard_mode() ? have == need : have >= need; }

hipError_t dev_malloc(void** out, size_t bytes) {
    if (!guard_mode()) {
        const hipError_t pe = hipMalloc(out, bytes);
        if (pe == hipSuccess) note_alloc(*out, bytes);
        return pe;
    }
    if (guard_mode() == 3) {
        hipError_t pe = hipMalloc(out, bytes);
        if (pe == hipSuccess) note_alloc(*out, bytes);
        if (pe == hipSuccess) pe = hipMemset(*out, 0xAB, bytes);
        if (pe == hipSuccess) pe = hipDeviceSynchronize();
        return pe;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    const size_t unit = (std::max<size_t>(bytes, 1) + 255) / 256 * 256;
    GuardedRange g = {};
    g.mapped_bytes = (unit + gran - 1) / gran * gran;
    g.va_bytes = g.mapped_bytes + 2 * gran;
    if ((e = hipMemAddressReserve(&g.va, g.va_bytes, gran, nullptr, 0)) != hipSuccess) return e;
    g.mapped = static_cast<char*>(g.va) + gran;
    if ((e = hipMemCreate(&g.handle, g.mapped_bytes, &prop, 0)) != hipSuccess) {
        (void)hipMemAddressFree(g.va, g.va_bytes);
        return e;
    }
    hipMemAccessDesc access = {};
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemMap(g.mapped, g.mapped_bytes, 0, g.handle, 0)) != hipSuccess ||
        (e = hipMemSetAccess(g.mapped, g.mapped_bytes, &access, 1)) != hipSuccess) {
        (void)hipMemUnmap(g.mapped, g.mapped_bytes);
        (void)hipMemRelease(g.handle);
        (void)hipMemAddressFree(g.va, g.va_bytes);
        return e;
    }
    g.user = guard_mode() == 2 ? static_cast<char*>(g.mapped) + (g.mapped_bytes - unit) : g.mapped;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        g_guarded.push_back(g);
    }
    *out = g.user;
    note_alloc(g.user, unit);
    // poison + synchronise: nothing may rely on fresh memory being zero, and a kernel queued right behind hipMemSetAccess
    // on a non-blocking stream was seen to read the new range before its mapping had settled (wrong values, no fault)
    e = hipMemset(g.mapped, 0xAB, g.mapped_bytes);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}

hipError_t dev_free(void* p) {
    if (p) note_free(p);
    if (!guard_mode() || guard_mode() == 3 || !p) return hipFree(p);
    GuardedRange g = {};
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        for (size_t i = 0; i < g_guarded.size(); ++i)
            if (g_guarded[i].user == p) {
                g = g_guarded[i];
                g_guarded.erase(g_guarded.begin() + i);
                break;
            }
    }
    if (!g.user) return hipFree(p);  // not one of ours (allocated before the switch was read: cannot happen; or foreign)
    hipError_t e = hipDeviceSynchronize();  // hipFree's implicit synchronisation
    if (e != hipSuccess) return e;
    // the address range stays reserved (and unmapped: a use after free faults as well); the physical memory goes back
    if ((e = hipMemUnmap(g.mapped, g.mapped_bytes)) != hipSuccess) return e;
    return hipMemRelease(g.handle);
}

struct Slot {
    hipStream_t stream = nullptr;
    void* dev = nullptr;      // staging slab of the host-pointer calls (owned by whoever holds the lease)
    size_t dev_bytes = 0;
    void* norm_ws = nullptr;  // workspace of hk_block_norm_dev on this stream (never shared with `dev`)
    size_t norm_ws_bytes = 0;
    void* aux = nullptr;      // on-demand planes + tables of the in-painting branch
    size_t aux_bytes = 0;
    // exchange buffer of hk_block_norm_split_comm_dev on THIS stream: sequences queued on different streams run concurrently on
    // the device (comm_mu only orders their queuing), so they must not share one
    double* comm_xchg = nullptr;
    size_t comm_xchg_doubles = 0;
    // Pinned words of the slot (PIN_BYTES of hipHostMalloc'd memory): small results and arguments travel through them,
    // never through the caller's pageable memory.  fail_host = the first word (the r2-mask failure counter).
    unsigned long long* fail_host = nullptr;
    static constexpr size_t PIN_NORM = 64, PIN_SUMS = 128, PIN_NORM_IN = 192, PIN_WORD = 256, PIN_COUNTS = 1024, PIN_BYTES = 16384;
    template <class T> T* pin(size_t off) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(fail_host) + off); }
    // Pinned staging ring of the host-pointer calls (stage_h2d / stage_d2h below): STAGE_N chunks of stage_chunk bytes.
    // A chunk is re-used once the copy that last used it has completed (stage_ev); a device-to-host chunk additionally
    // carries the unpacking into the caller's array that follows its copy.
    static constexpr int STAGE_N = 4;
    struct StageOut {
        char* h_dst = nullptr;
        size_t h_pitch = 0, row_bytes = 0, c_pitch = 0;
        size_t rows = 0;
    };
    char* stage = nullptr;
    size_t stage_chunk = 0;
    hipEvent_t stage_ev[STAGE_N] = {nullptr, nullptr, nullptr, nullptr};
    int stage_state[STAGE_N] = {0, 0, 0, 0};  // 0 free, 1 host-to-device copy queued, 2 device-to-host copy queued (unpack follows)
    StageOut stage_out[STAGE_N];
    int stage_next = 0;
    // job / plane tables of the batched device entry points: a small ring of (pinned host, device) buffer pairs; entry i is
    // re-used once the upload recorded in tbl_ev[i] has been made (the device side is ordered by the stream itself)
    static constexpr int TBL_RING = 4;
    void* tbl_host[TBL_RING] = {nullptr, nullptr, nullptr, nullptr};
    void* tbl_dev[TBL_RING] = {nullptr, nullptr, nullptr, nullptr};
    size_t tbl_bytes[TBL_RING] = {0, 0, 0, 0};
    hipEvent_t tbl_ev[TBL_RING] = {nullptr, nullptr, nullptr, nullptr};
    int tbl_next = 0;
    // gain-offset with the r2 mask: the wave-rows the certificate build leaves to the list launch (FitArgs::open_rows), one bit each
    unsigned* open_rows = nullptr;
    size_t open_rows_words = 0;
    bool direct_pending = false;  // copies queued straight from / to page-locked CALLER arrays since the last stream synchronisation
    bool busy = false;         // leased by a host-pointer call (SlotLease)
    int dev_inflight = 0;      // device-job entry points currently queuing on this stream (DevEnter): a lease waits for them
    bool dev_touched = false;  // device-resident jobs were queued on this stream since the last lease drained it
};

constexpr int64_t ROW_ALIGN = 64;  // device rows padded to 64 elements (256 B)

}  // namespace

struct hk_ctx {
    int device = 0;
    std::vector<Slot> slots;
    std::mutex mu;
    std::condition_variable cv;
    int xcd_remap = 0;
    // the last HOST-POINTER gain-offset call with a threshold found pixels failing the r2 mask: real imagery usually does, block
    // after block, so the next call lets its first pass leave what the in-painting reads (offsets + source flags, 5 bytes per
    // pixel of stores) instead of running it again when the count comes back non-zero.  Either choice gives the same results.
    // (Rounds 3-5 also chose the kernel BUILD by it -- certificate-only or complete -- with a back-off and, on the device-job
    // path, an expiry count; since round 6 the certificate build always runs first and hands the wave-rows it cannot settle to
    // a list launch, and a device-resident job says by carrying `scratch` that it wants the in-painting's inputs left there.)
    std::atomic<int> expect_r2_failures{0};
    // RCCL communicator of the one data-path collective (hk_comm_init; the split-block statistics)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    std::mutex comm_mu;          // collectives of one communicator are queued in one order on every rank
    // hk_memcpy_h2d / hk_memcpy_d2h / hk_selftest: a stream + pinned staging ring of their own, one caller at a time
    Slot xfer;
    std::mutex xfer_mu;
};

struct hk_event {
    hipEvent_t ev;
};

namespace {

void stage_abandon(Slot& s);  // (host staging, below)

struct SlotLease {
    hk_ctx* ctx;
    int idx;
    // Host-pointer calls and device-resident jobs share the pooled streams and their scratch (norm_ws, aux).  The rule "do
    // not mix them on one context concurrently" is enforced here instead of being left to the caller: a lease prefers a
    // stream no device job has touched; if it has to take one that was, it drains that stream first (the jobs queued on it
    // are complete before the slot's scratch is re-used), and a device-job call waits while its stream is leased
    // (dev_slot_enter below).
    SlotLease(hk_ctx* c) : ctx(c), idx(-1) {
        bool drain = false;
        {
            std::unique_lock<std::mutex> lk(ctx->mu);
            ctx->cv.wait(lk, [&] {
                int any = -1;
                for (size_t i = 0; i < ctx->slots.size(); ++i) {
                    if (ctx->slots[i].busy || ctx->slots[i].dev_inflight > 0) continue;
                    if (!ctx->slots[i].dev_touched) {
                        idx = (int)i;
                        return true;
                    }
                    if (any < 0) any = (int)i;
                }
                idx = any;
                return any >= 0;
            });
            ctx->slots[idx].busy = true;
            drain = ctx->slots[idx].dev_touched;
            ctx->slots[idx].dev_touched = false;
        }
        if (drain) (void)hipStreamSynchronize(ctx->slots[idx].stream);
    }
    ~SlotLease() {
        // a call that ends through stage_finish() leaves no chunk queued; one that returned an error earlier does -- see stage_abandon
        stage_abandon(ctx->slots[idx]);
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->slots[idx].busy = false;
        }
        ctx->cv.notify_all();
    }
    Slot& slot() { return ctx->slots[idx]; }
};

// A device-resident job is about to be queued on pooled stream `stream`: wait for a host-pointer call that holds it, and keep
// the slot from being leased until the entry point has finished queuing (it uses the slot's scratch -- norm_ws, aux, the table
// ring -- after this check; a lease taken in between would drain an still-empty stream and use the same scratch).
struct DevEnter {
    hk_ctx* ctx;
    int stream;
    DevEnter(hk_ctx* c, int st) : ctx(c), stream(st) {
        std::unique_lock<std::mutex> lk(ctx->mu);
        ctx->cv.wait(lk, [&] { return !ctx->slots[stream].busy; });
        ctx->slots[stream].dev_touched = true;
        ++ctx->slots[stream].dev_inflight;
    }
    ~DevEnter() {
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            --ctx->slots[stream].dev_inflight;
        }
        ctx->cv.notify_all();
    }
    DevEnter(const DevEnter&) = delete;
    DevEnter& operator=(const DevEnter&) = delete;
};

// RCCL, opened on first use: the library itself stays loadable (and its CPU-side tests runnable) where librccl is absent
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce && api.GetErrorString;
    });
    return api;
}
#define HK_RCCL(expr)                                                                                         \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess)                                                                                \
            return fail(HK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

int ensure_dev(Slot& s, size_t bytes) {
    if (fits(s.dev_bytes, bytes)) return HK_OK;
    if (s.dev) {
        HK_HIP(hipStreamSynchronize(s.stream));
        HK_HIP(dev_free(s.dev));
        s.dev = nullptr;
        s.dev_bytes = 0;
    }
    const size_t want = guard_mode() ? bytes : bytes + bytes / 8;  // (guarded: the buffer ends where its last user's does)
    if (dev_malloc(&s.dev, want) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", want);
    s.dev_bytes = want;
    return HK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host staging.  BASELINE.json north_star: "blocks stream ... into pinned host buffers and hipMemcpyAsync to HBM".  The HIP
// runtime is only ever handed page-locked memory: the library's own ring (pack on the host -> ONE contiguous hipMemcpyAsync
// per chunk -> pitched layout on the device, and the reverse), or the caller's arrays when those are page-locked already
// (hk_host_alloc / hk_host_register: direct, fully asynchronous).  The runtime's pageable paths -- pinning caller pages on
// the fly, rect copies into unaligned few-byte rows of a numpy buffer -- are not part of the product.
constexpr size_t STAGE_CHUNK_MIN = 8u << 20;

// Page-locked host ranges this library made (hk_host_alloc, hk_host_register), process-wide: base -> bytes.  A caller array
// is handed to the runtime's copy engines directly only if ONE of these ranges holds all of it -- every row, every byte between
// the first and the last.  (Rounds 3-4 asked the runtime about the array's first and last byte: a strided view whose ends lie in
// two different registered ranges with pageable memory between them passed, and the runtime pinned the gap on the fly -- the
// path of profiles/r04_abort_caught.txt.)  Memory page-locked by other means is unknown here and goes through the staging ring.
struct PinnedRanges {
    std::mutex mu;
    std::map<uintptr_t, size_t> r;
    void add(const void* p, size_t n) {
        std::lock_guard<std::mutex> lk(mu);
        r[reinterpret_cast<uintptr_t>(p)] = n;
    }
    void remove(const void* p) {
        std::lock_guard<std::mutex> lk(mu);
        r.erase(reinterpret_cast<uintptr_t>(p));
    }
    bool covers(const void* p, size_t n) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        std::lock_guard<std::mutex> lk(mu);
        auto it = r.upper_bound(a);
        if (it == r.begin()) return false;
        --it;
        return a >= it->first && n <= it->second && a - it->first <= it->second - n;
    }
};
PinnedRanges& pinned_ranges() {
    static PinnedRanges g;
    return g;
}
// copies handed to the runtime straight from / to caller memory, and chunks that went through the staging ring (hk_debug_staging_counters)
std::atomic<unsigned long long> g_direct_copies{0}, g_staged_chunks{0};
std::atomic<int> g_fail_after_d2h{0};  // fault injection of the test-suite (hk_debug_fail_after_d2h)

// is [p, p + bytes) inside one page-locked range of this library?
bool host_is_pinned(const void* p, size_t bytes) { return bytes > 0 && pinned_ranges().covers(p, bytes); }

// HK_ASSERT_PINNED=1 (the test-suite sets it): before a caller pointer goes to hipMemcpy*Async directly, ask the runtime about
// EVERY page of every row; a page it does not know as host memory fails the call instead of being pinned on the fly.
bool assert_pinned_on() {
    static const bool on = [] { const char* e = getenv("HK_ASSERT_PINNED"); return e && atoi(e) != 0; }();
    return on;
}
int check_rows_pinned(const void* p, size_t pitch, size_t row_bytes, size_t rows) {
    const char* base = static_cast<const char*>(p);
    for (size_t r = 0; r < rows; ++r) {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(base + r * pitch), a1 = a0 + row_bytes - 1;
        for (uintptr_t pg = a0 & ~(uintptr_t)4095; pg <= a1; pg += 4096) {
            hipPointerAttribute_t attr;
            const void* q = reinterpret_cast<const void*>(pg < a0 ? a0 : pg);
            const bool ok = hipPointerGetAttributes(&attr, q) == hipSuccess && attr.type == hipMemoryTypeHost;
            (void)hipGetLastError();
            if (!ok) return fail(HK_ERR_ARG, "HK_ASSERT_PINNED: host address %p (row %zu) is not page-locked but was about to be copied directly", q, r);
        }
    }
    return HK_OK;
}

int ensure_pin(Slot& s) {
    if (s.fail_host) return HK_OK;
    if (hipHostMalloc(reinterpret_cast<void**>(&s.fail_host), Slot::PIN_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        s.fail_host = nullptr;
        return fail(HK_ERR_NOMEM, "hipHostMalloc(%zu) failed", (size_t)Slot::PIN_BYTES);
    }
    memset(s.fail_host, 0, Slot::PIN_BYTES);
    return HK_OK;
}

// the copy (and unpacking) that last used chunk `i` is complete
int stage_settle(Slot& s, int i) {
    if (s.stage_state[i] == 0) return HK_OK;
    HK_HIP(hipEventSynchronize(s.stage_ev[i]));
    if (s.stage_state[i] == 2) {
        const Slot::StageOut& o = s.stage_out[i];
        const char* c = s.stage + (size_t)i * s.stage_chunk;
        if (o.h_pitch == o.row_bytes && o.c_pitch == o.row_bytes) memcpy(o.h_dst, c, o.rows * o.row_bytes);
        else
            for (size_t r = 0; r < o.rows; ++r) memcpy(o.h_dst + r * o.h_pitch, c + r * o.c_pitch, o.row_bytes);
    }
    s.stage_state[i] = 0;
    return HK_OK;
}

// every queued staging copy of the slot has completed and has been unpacked (the caller's output arrays are final)
int stage_drain(Slot& s) {
    if (!s.stage) return HK_OK;
    for (int k = 0; k < Slot::STAGE_N; ++k) {
        const int rc = stage_settle(s, (s.stage_next + k) % Slot::STAGE_N);  // oldest first
        if (rc) return rc;
    }
    return HK_OK;
}

int ensure_stage(Slot& s, size_t min_chunk) {
    // HK_STAGE_CHUNK_KB: chunk size for tests of the chunked paths (default 8 MB; never below one device row)
    static const size_t chunk_min = [] {
        const char* e = getenv("HK_STAGE_CHUNK_KB");
        return (e && atol(e) > 0) ? (size_t)atol(e) << 10 : STAGE_CHUNK_MIN;
    }();
    const size_t want = std::max(min_chunk, chunk_min);
    if (s.stage && s.stage_chunk >= want) return HK_OK;
    int rc = stage_drain(s);
    if (rc) return rc;
    if (s.stage) HK_HIP(hipHostFree(s.stage));
    s.stage = nullptr, s.stage_chunk = 0;
    const size_t chunk = (want + 4095) / 4096 * 4096;
    if (hipHostMalloc(reinterpret_cast<void**>(&s.stage), chunk * Slot::STAGE_N, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        s.stage = nullptr;
        return fail(HK_ERR_NOMEM, "hipHostMalloc(%zu) failed", chunk * Slot::STAGE_N);
    }
    s.stage_chunk = chunk;
    for (int i = 0; i < Slot::STAGE_N; ++i)
        if (!s.stage_ev[i]) HK_HIP(hipEventCreateWithFlags(&s.stage_ev[i], hipEventDisableTiming));
    return HK_OK;
}

int stage_take(Slot& s, int* chunk) {
    g_staged_chunks.fetch_add(1, std::memory_order_relaxed);
    const int i = s.stage_next;
    s.stage_next = (s.stage_next + 1) % Slot::STAGE_N;
    const int rc = stage_settle(s, i);
    *chunk = i;
    return rc;
}

// `rows` rows of `row_bytes` from the host (rows h_pitch bytes apart) to the device (rows d_pitch bytes apart), queued on the
// slot's stream.  The host side may be released when the call returns (pageable memory: it has been packed; page-locked
// memory: the caller keeps it until the stream has been synchronised, as the entry points do before they return).
int stage_h2d(Slot& s, void* d_dst, size_t d_pitch, const void* h_src, size_t h_pitch, size_t row_bytes, size_t rows) {
    if (rows == 0 || row_bytes == 0) return HK_OK;
    if (rows == 1) h_pitch = d_pitch = row_bytes;
    const bool flat = h_pitch == row_bytes && d_pitch == row_bytes;
    if (host_is_pinned(h_src, (rows - 1) * h_pitch + row_bytes)) {
        if (assert_pinned_on()) {
            const int rc = check_rows_pinned(h_src, h_pitch, row_bytes, rows);
            if (rc) return rc;
        }
        g_direct_copies.fetch_add(1, std::memory_order_relaxed);
        s.direct_pending = true;
        if (flat) HK_HIP(hipMemcpyAsync(d_dst, h_src, rows * row_bytes, hipMemcpyHostToDevice, s.stream));
        else HK_HIP(hipMemcpy2DAsync(d_dst, d_pitch, h_src, h_pitch, row_bytes, rows, hipMemcpyHostToDevice, s.stream));
        return HK_OK;
    }
    int rc = ensure_stage(s, flat ? 0 : d_pitch);
    if (rc) return rc;
    const char* h = static_cast<const char*>(h_src);
    char* d = static_cast<char*>(d_dst);
    if (flat) {  // a byte stream
        const size_t total = rows * row_bytes;
        for (size_t off = 0; off < total; off += s.stage_chunk) {
            const size_t n = std::min(s.stage_chunk, total - off);
            int i;
            if ((rc = stage_take(s, &i))) return rc;
            char* c = s.stage + (size_t)i * s.stage_chunk;
            memcpy(c, h + off, n);
            HK_HIP(hipMemcpyAsync(d + off, c, n, hipMemcpyHostToDevice, s.stream));
            HK_HIP(hipEventRecord(s.stage_ev[i], s.stream));
            s.stage_state[i] = 1;
        }
        return HK_OK;
    }
    // packed at the DEVICE pitch: the chunk maps onto the device rows with one contiguous copy (the padding between two
    // rows travels along; it belongs to the plane and nobody reads it)
    const size_t per = std::max<size_t>(s.stage_chunk / d_pitch, 1);
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t n = std::min(per, rows - r0);
        int i;
        if ((rc = stage_take(s, &i))) return rc;
        char* c = s.stage + (size_t)i * s.stage_chunk;
        for (size_t r = 0; r < n; ++r) memcpy(c + r * d_pitch, h + (r0 + r) * h_pitch, row_bytes);
        HK_HIP(hipMemcpyAsync(d + r0 * d_pitch, c, (n - 1) * d_pitch + row_bytes, hipMemcpyHostToDevice, s.stream));
        HK_HIP(hipEventRecord(s.stage_ev[i], s.stream));
        s.stage_state[i] = 1;
    }
    return HK_OK;
}

// the reverse; the caller's array is final after stage_drain() (pageable) / after the stream has been synchronised (pinned)
int stage_d2h(Slot& s, void* h_dst, size_t h_pitch, const void* d_src, size_t d_pitch, size_t row_bytes, size_t rows) {
    if (rows == 0 || row_bytes == 0) return HK_OK;
    if (rows == 1) h_pitch = d_pitch = row_bytes;
    const bool flat = h_pitch == row_bytes && d_pitch == row_bytes;
    if (host_is_pinned(h_dst, (rows - 1) * h_pitch + row_bytes)) {
        if (assert_pinned_on()) {
            const int rc = check_rows_pinned(h_dst, h_pitch, row_bytes, rows);
            if (rc) return rc;
        }
        g_direct_copies.fetch_add(1, std::memory_order_relaxed);
        s.direct_pending = true;
        if (flat) HK_HIP(hipMemcpyAsync(h_dst, d_src, rows * row_bytes, hipMemcpyDeviceToHost, s.stream));
        else HK_HIP(hipMemcpy2DAsync(h_dst, h_pitch, d_src, d_pitch, row_bytes, rows, hipMemcpyDeviceToHost, s.stream));
        return HK_OK;
    }
    int rc = ensure_stage(s, flat ? 0 : d_pitch);
    if (rc) return rc;
    char* h = static_cast<char*>(h_dst);
    const char* d = static_cast<const char*>(d_src);
    if (flat) {
        const size_t total = rows * row_bytes;
        for (size_t off = 0; off < total; off += s.stage_chunk) {
            const size_t n = std::min(s.stage_chunk, total - off);
            int i;
            if ((rc = stage_take(s, &i))) return rc;
            HK_HIP(hipMemcpyAsync(s.stage + (size_t)i * s.stage_chunk, d + off, n, hipMemcpyDeviceToHost, s.stream));
            HK_HIP(hipEventRecord(s.stage_ev[i], s.stream));
            s.stage_state[i] = 2;
            s.stage_out[i].h_dst = h + off, s.stage_out[i].h_pitch = n, s.stage_out[i].row_bytes = n;
            s.stage_out[i].c_pitch = n, s.stage_out[i].rows = 1;
        }
        return HK_OK;
    }
    const size_t per = std::max<size_t>(s.stage_chunk / d_pitch, 1);
    for (size_t r0 = 0; r0 < rows; r0 += per) {
        const size_t n = std::min(per, rows - r0);
        int i;
        if ((rc = stage_take(s, &i))) return rc;
        HK_HIP(hipMemcpyAsync(s.stage + (size_t)i * s.stage_chunk, d + r0 * d_pitch, (n - 1) * d_pitch + row_bytes,
                              hipMemcpyDeviceToHost, s.stream));
        HK_HIP(hipEventRecord(s.stage_ev[i], s.stream));
        s.stage_state[i] = 2;
        s.stage_out[i].h_dst = h + r0 * h_pitch, s.stage_out[i].h_pitch = h_pitch, s.stage_out[i].row_bytes = row_bytes;
        s.stage_out[i].c_pitch = d_pitch, s.stage_out[i].rows = n;
    }
    return HK_OK;
}

// A host-pointer call is ending WITHOUT stage_finish (an error return somewhere behind a stage_d2h): its queued chunks still name
// the caller's output arrays, which the caller may free as soon as it sees the error -- the next call on this slot must not unpack
// into them.  Let the copies that were queued complete (they write the ring, never the caller) and forget them.  The same goes for
// DIRECT copies (page-locked caller arrays, the RasterFuse default): they are still in flight on the caller's memory, which the
// caller may unregister and free once it has the error -- the stream is drained before the call returns (round-5 advisor finding).
void stage_abandon(Slot& s) {
    bool any = s.direct_pending;
    for (int i = 0; i < Slot::STAGE_N; ++i) any |= s.stage_state[i] != 0;
    if (!any) return;
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    (void)hipGetLastError();
    for (int i = 0; i < Slot::STAGE_N; ++i) s.stage_state[i] = 0;
    s.direct_pending = false;
}

// end of a host-pointer call: everything queued on the slot's stream has run and every output array is final
int stage_finish(Slot& s) {
    const int rc = stage_drain(s);
    if (rc) return rc;
    HK_HIP(hipStreamSynchronize(s.stream));
    s.direct_pending = false;
    return HK_OK;
}

void slot_release(Slot& s) {
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.dev) (void)dev_free(s.dev);
    if (s.norm_ws) (void)dev_free(s.norm_ws);
    if (s.aux) (void)dev_free(s.aux);
    if (s.comm_xchg) (void)dev_free(s.comm_xchg);
    if (s.open_rows) (void)dev_free(s.open_rows);
    if (s.fail_host) (void)hipHostFree(s.fail_host);
    if (s.stage) (void)hipHostFree(s.stage);
    for (int i = 0; i < Slot::STAGE_N; ++i)
        if (s.stage_ev[i]) (void)hipEventDestroy(s.stage_ev[i]);
    for (int i = 0; i < Slot::TBL_RING; ++i) {
        if (s.tbl_host[i]) (void)hipHostFree(s.tbl_host[i]);
        if (s.tbl_dev[i]) (void)dev_free(s.tbl_dev[i]);
        if (s.tbl_ev[i]) (void)hipEventDestroy(s.tbl_ev[i]);
    }
    if (s.stream) (void)hipStreamDestroy(s.stream);
    s = Slot();
}

// rasterio.enums.Resampling values with a device kernel (hk_resample.hip)
bool resampling_built(int m) { return (m >= 0 && m <= 6) || (m >= 8 && m <= 14); }  // all of GRA_* but gauss (7: not a warp method)

bool needs_r2(const hk_fit_desc* d) {
    return d->find_r2 || (d->model == HK_MODEL_GAIN_OFFSET && d->has_r2_thresh);
}

int validate_desc(const hk_fit_desc* d) {
    if (!d) return fail(HK_ERR_ARG, "desc is NULL");
    if (d->model < 0 || d->model > 2) return fail(HK_ERR_ARG, "unknown model %d", d->model);
    // homonim/utils.py:121-132
    if (d->kh < 1 || d->kw < 1) return fail(HK_ERR_ARG, "`kernel_shape` must be a minimum of one in both dimensions.");
    if ((d->kh & 1) == 0 || (d->kw & 1) == 0) return fail(HK_ERR_ARG, "`kernel_shape` must be odd in both dimensions.");
    if (d->model == HK_MODEL_GAIN_OFFSET && d->kh * d->kw < 2)
        return fail(HK_ERR_ARG, "`kernel_shape` area should contain at least 2 elements for the gain-offset model.");
    if (d->kh > 255) return fail(HK_ERR_UNSUPPORTED, "kernel height %d > 255 not supported", d->kh);
    if (hk::overlap_lanes_for(d->kw / 2) > 24) return fail(HK_ERR_UNSUPPORTED, "kernel width %d too large", d->kw);
    for (int m : {d->src_nodata_mode, d->ref_nodata_mode})
        if (m < 0 || m > 2) return fail(HK_ERR_ARG, "bad nodata mode %d", m);
    return HK_OK;
}

// The reference decides `(1 - f32(f64(ssres / sstot))) > thresh` (kernel_model.py:212-213,363).  Returns a float64 factor
// c such that, for sstot > 0, `ssres < c * sstot` PROVES that decision true: c sits 2^-40 (relative) below the
// float32 rounding boundary of the largest quotient that still passes, which swallows the 2^-53 errors of the float64
// division and of the comparison product.  -inf = nothing can be certified (threshold >= 1 or NaN).
double r2_pass_scale(float thresh) {
    if (!(thresh < 1.0f)) return -INFINITY;
    // largest float q with (1.0f - q) > thresh: the predicate is monotone non-increasing in q
    auto key = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
    auto unkey = [](uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; };
    uint32_t lo = key(-FLT_MAX), hi = key(FLT_MAX);  // pred(lo) is true for every thresh < 1
    auto pred = [&](uint32_t k) { volatile float d = 1.0f - unkey(k); return d > thresh; };
    if (!pred(lo)) return -INFINITY;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo + 1) / 2;
        if (pred(mid)) lo = mid; else hi = mid - 1;
    }
    const float q = unkey(lo);
    const float qn = nextafterf(q, INFINITY);
    const double boundary = std::isinf(qn) ? (double)q : 0.5 * ((double)q + (double)qn);
    if (!(boundary > 0.0)) return -INFINITY;
    return boundary * (1.0 - 0x1p-40);
}

// The mirror image: for sstot > 0, `ssres > c_hi * sstot` PROVES the decision false (c_hi sits 2^-40 above the same
// rounding boundary).  +inf = failure is never certified (degenerate thresholds): the division decides.
double r2_fail_above(float thresh) {
    const double c = r2_pass_scale(thresh);
    if (!(c > 0.0)) return INFINITY;
    return c / (1.0 - 0x1p-40) * (1.0 + 0x1p-40);
}

// kappa >= 1 - c' with c' = c * (1 - 2^-50) (so that `ssres < c' * sstot` in real arithmetic implies the float64
// comparison against fl(c * sstot)), clamped to >= 0 and rounded up to float32.  +inf = nothing can be certified.
float r2_fail_scale(float thresh) {
    const double c = r2_pass_scale(thresh);
    if (!(c > 0.0)) return INFINITY;
    double k = 1.0 - c * (1.0 - 0x1p-50);
    if (k < 0.0) k = 0.0;
    float kf = (float)k;
    if ((double)kf < k + 0x1p-60) kf = nextafterf(kf, INFINITY);
    return kf;
}

// kappa_f <= 1 - c_hi * (1 + 2^-50) with c_hi = r2_fail_above(): `g^2 den < kappa_f * sstot - (rounding slack)` then gives
// ssres > c_hi * sstot in real arithmetic with room for the float64 comparison, i.e. the reference's decision is False
// (PROOFS.md appendix A, the fail side).  Rounded DOWN to float32; -inf = failure is never certified that way.
float r2_failcert_scale(float thresh) {
    const double c_hi = r2_fail_above(thresh);
    if (!(c_hi > 0.0) || std::isinf(c_hi)) return -INFINITY;
    const double k = 1.0 - c_hi * (1.0 + 0x1p-50);
    if (!(k > 0.0)) return -INFINITY;
    float kf = (float)k;
    if ((double)kf > k - 0x1p-60) kf = nextafterf(kf, -INFINITY);
    return kf;
}

void fill_args(hk::FitArgs& a, const hk_fit_desc* d, int xcd_remap) {
    a.rh = d->kh / 2;
    a.rw = d->kw / 2;
    a.overlap_lanes = hk::overlap_lanes_for(a.rw);
    a.src_nd_mode = d->src_nodata_mode;
    a.ref_nd_mode = d->ref_nodata_mode;
    a.src_nodata = d->src_nodata;
    a.ref_nodata = d->ref_nodata;
    a.has_thresh = (d->model == HK_MODEL_GAIN_OFFSET) ? d->has_r2_thresh : 0;
    a.r2_thresh = d->r2_thresh;
    a.r2_fail_scale = a.has_thresh ? r2_fail_scale(d->r2_thresh) : INFINITY;
    a.r2_failcert_scale = a.has_thresh ? r2_failcert_scale(d->r2_thresh) : -INFINITY;
    a.r2_pass_below = a.has_thresh ? r2_pass_scale(d->r2_thresh) : -INFINITY;
    a.r2_fail_above = a.has_thresh ? r2_fail_above(d->r2_thresh) : INFINITY;
    a.n_full = (float)(d->kh * d->kw);
    a.nd_full = (double)(d->kh * d->kw);
    a.inv_n_full = 1.0 / (double)(d->kh * d->kw);
    a.force_general = getenv("HK_FORCE_GENERAL") ? atoi(getenv("HK_FORCE_GENERAL")) : 0;
    // ring mode (hk_fit_kernel.h): full LDS ring while it leaves room for >= 11 waves per CU (kh <= 5), centre-only ring up
    // to kh = 39 (1 KB per wave and row of the half-height; 33 - 39 rows: 3 - 13 % faster than re-loading, from 41 rows
    // slower -- headline workload, round 5), everything re-loaded beyond -- that path exists
    // for kernels from 9 wide only (launch_rw), narrower ones keep the centre ring whatever their height (kh <= 255: 128 KB)
    a.use_ring = (d->kh <= 5 && d->kw <= 7) ? 1 : ((d->kh <= 39 || d->kw <= 7) ? 2 : 0);
    // The memory-bound builds (no R2) prefer the full ring well beyond that: re-loading the leaving rows costs them more than
    // the waves the ring displaces -- gain 7x7 / 9x9 / 11x11 / 15x15 at 16384^2 x 4: 3.25 / 3.40 / 3.45 / 3.95 -> 2.54 / 2.58 /
    // 2.77 / 3.64 ms; gain-blk-offset (more arithmetic per pixel) only up to 7x7 (fit + statistics 5.11 -> 4.61 ms; 9x9 equal,
    // 15x15 +21 %)
    if (!needs_r2(d) && d->kw <= 15) {
        if (d->model == HK_MODEL_GAIN && d->kh <= 15) a.use_ring = 1;
        if (d->model == HK_MODEL_GAIN_BLK_OFFSET && d->kh <= 7 && d->kw <= 7) a.use_ring = 1;
        // Round 3, the SPLIT ring (rh rows in registers + rh + 1 in LDS: no re-load, 16 instead of 30 KB of LDS at 15x15):
        // gain 11x11 / 13x13 / 15x15 at 16384^2 x 4: 2.77 / 3.08 / 3.47 -> 2.70 / 2.86 / 3.07 ms (7x7 and 9x9 stay with the
        // full ring: 2.36 / 2.45 against 2.48 / 2.72); gain-blk-offset incl. statistics 9x9 / 11x11: 5.23 / 5.38 -> 4.95 /
        // 5.27 ms -- but 13x13 / 15x15 get SLOWER (5.37 / 5.35 -> 5.57 / 5.99): with its float64 normalisation and quotient
        // per pixel and the 15-wide horizontal sums that kernel is bound by VALU issue, not by the 8 B per pixel the
        // re-load moves (profiles/r03_sweep_ring.txt).
        if (d->kw >= 5) {
            if (d->model == HK_MODEL_GAIN && d->kh >= 11 && d->kh <= 15) a.use_ring = 3;
            if (d->model == HK_MODEL_GAIN_BLK_OFFSET && d->kh >= 9 && d->kh <= 11) a.use_ring = 3;
        }
    }
    // ... and, in their lock-step workgroups (hk_fit_kernel.h WPB), short uniform segments: 32 rows instead of 64 / 128 + 32
    // (gain 5x5: configs[1] 0.676 -> 0.625 ms, 16384^2 x 4 2.65 -> 2.49 ms on one box; 7x7 2.54 -> 2.47; gain-blk-offset 5x5
    // -0.5 %).  16 rows are as good or better at 5x5 but load a quarter more rows; taller kernels keep the default.
    a.seg_rows_pref = 0;
    if (!needs_r2(d) && a.use_ring == 1 && d->model != HK_MODEL_GAIN_OFFSET &&
        (d->kh <= 5 || (d->model == HK_MODEL_GAIN && d->kh <= 7)))
        a.seg_rows_pref = 32;
    if (const char* e = getenv("HK_USE_RING")) {  // testing hook
        const int m = atoi(e);
        const bool mem_bound = d->model != HK_MODEL_GAIN_OFFSET && !needs_r2(d);
        if (m == 1 && d->kh <= 63 && (d->kw <= 7 || (mem_bound && d->kw <= 15))) a.use_ring = 1;
        if (m == 2 && d->kh <= 127) a.use_ring = 2;
        if (m == 3 && mem_bound && d->kh >= 7 && d->kh / 2 <= hk::split_ring_rows(d->model) && d->kw >= 5 && d->kw <= 15) a.use_ring = 3;
        if (m == 0 && d->kw >= 9) a.use_ring = 0;
    }
    a.xcd_remap = xcd_remap;
    // Fewer resident waves for the dense `gain` kernel on large rasters: at 8 KB of LDS per wave 20 waves per CU stream their rows
    // at once and the HBM answers with ~5.0 TB/s; 4 KB of unused LDS per wave leave 12 (three lock-step workgroups per CU) and it
    // answers with 5.4: gain 5x5 at 16384^2 x 4 2.58 -> 2.37 ms, configs[1] 0.631 -> 0.605 ms, 3x3 -2.5 % (profiles/r03_lds_pad.txt).
    // Not for the general (nodata) build -- its registers keep it at that occupancy already --, not for 7x7 (12 KB ring), not for
    // launches of less than ~128 M pixels (a 4-band 4096^2 tile: +3 %); the VALU-bound builds want every wave (headline +8 % at 16
    // -> 14 waves).  -1 = decided by fill_grid() from the job's size.
    a.lds_pad = 0;
    if (d->model == HK_MODEL_GAIN && !needs_r2(d) && d->kh <= 5 && d->kw <= 7 && d->src_nodata_mode == HK_NODATA_NONE &&
        d->ref_nodata_mode == HK_NODATA_NONE && !a.force_general)
        a.lds_pad = -1;
    a.out_y0 = 0, a.out_y1 = a.height, a.out_x0 = 0, a.out_x1 = a.width;  // store window: the whole job (callers narrow it)
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Units of one launch: (segment, band, strip), one wave each.  A wave re-reads 2*rh priming rows, so long segments waste
// less; but the launch ends when its last wave does, so the final segments are short.  Large rasters therefore get
// `big` rows per segment for most of the height and `tail` rows for the last ~1.25 "generations" of resident waves
// (dispatched last, segment-major order); small rasters and an explicit `seg_rows` use one size.
void fill_grid(hk::FitArgs& a, int seg_rows) {
    const int out_w = (hk::WAVE - 2 * a.overlap_lanes) * hk::PX;
    const int kh = 2 * a.rh + 1;
    a.n_strips = (a.width + out_w - 1) / out_w;
    if (seg_rows <= 0 && a.seg_rows_pref > 0) seg_rows = a.seg_rows_pref;  // the build's own preference (fill_args)
    const int uniform = seg_rows > 0 ? seg_rows : (kh <= 5 ? 64 : (kh <= 9 ? 128 : 256));
    a.seg_rows = uniform < a.height ? uniform : a.height;
    a.seg_rows_tail = a.seg_rows;
    a.n_segs = (a.height + a.seg_rows - 1) / a.seg_rows;
    a.n_segs_big = a.n_segs;
    const long long slots = (long long)env_int("HK_WAVE_SLOTS", 256 * 12);  // resident waves of the device (3 per SIMD)
    const long long per_row_band = (long long)a.n_strips * a.n_bands;        // units per segment row
    if (seg_rows <= 0 && per_row_band * a.n_segs >= 6 * slots) {
        const int big = env_int("HK_SEG_BIG", 2 * uniform), tail = env_int("HK_SEG_TAIL", uniform / 2);
        const double gens = 1.25;
        // image rows whose big-segment units make up `gens` generations of resident waves
        long long tail_rows = (long long)(gens * (double)slots / (double)per_row_band * big);
        tail_rows = (tail_rows + tail - 1) / tail * tail;
        if (big > 0 && tail > 0 && tail_rows < a.height - big) {
            const int n_big = (int)((a.height - tail_rows) / big);
            const int rest = a.height - n_big * big;
            a.seg_rows = big;
            a.seg_rows_tail = tail;
            a.n_segs_big = n_big;
            a.n_segs = n_big + (rest + tail - 1) / tail;
        }
    }
    a.total_units = a.n_strips * a.n_segs * a.n_bands;
    if (a.lds_pad < 0) a.lds_pad = ((long long)a.height * a.width * a.n_bands >= (128ll << 20)) ? 4096 : 0;
}

// kernel_model.py:364-371 for ONE band whose first pass counted failing pixels: in-paint the offsets of the failing
// pixels from the passing ones (restated GDALFillNodata) and run the fit again with `offset_in`, which recomputes their
// gains and re-applies.  `a` is the first pass's argument block (n_bands == 1).
// scratch of the in-painting branch: [offset | column tables] (rounds 1-4 kept a `filled` plane and room for gain / r2 in front: the
// targets are filled in place since round 5)
static int ensure_inpaint_scratch(Slot& sl, size_t plane, int height, long long stride) {
    const size_t need = plane + hk::inpaint_workspace_bytes(height, stride);
    if (!fits(sl.aux_bytes, need)) {
        if (sl.aux) {
            HK_HIP(hipStreamSynchronize(sl.stream));
            HK_HIP(dev_free(sl.aux));
        }
        sl.aux = nullptr, sl.aux_bytes = 0;
        if (dev_malloc(&sl.aux, need) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", need);
        sl.aux_bytes = need;
    }
    return HK_OK;
}

// gain-offset with the r2 mask when nothing but the corrected block (and, from scratch, the offsets) is asked for and the failures
// are counted: the CERTIFICATE build runs first (118 - 126 registers, four waves per SIMD: the division-free float32 certificate of
// PROOFS.md appendix A settles a wave-row whose every valid pixel certainly passes) and marks the wave-rows it cannot settle in
// a bit plane; the LIST launch -- the complete build on a persistent grid, one run of marked rows per wave at a time -- follows on
// the same stream and does those rows with the reference's own R2 expression.  Together they write every row once and count every
// failing pixel once; an empty bit plane (clean rasters) costs the list launch a few microseconds.  Rounds 3 - 5 instead voided the
// whole band on the first open wave-row (HK_COUNT_RETRY in its counter), ran it again with the complete build and remembered, with
// a back-off and an expiry count, which build to start the next launches with.  (The certificate has builds for the full and the
// centre ring, its constants assume window counts below 2^16, and the in-painting's source flags do not fit its registers: other
// shapes, R2 / gain output and launches that leave the in-painting's inputs run the complete build over the whole grid.)
static_assert(HK_COUNT_RETRY == hk::FIT_RETRY_BIT, "public and kernel-side re-run bits differ");
static bool cert_list_eligible(const hk::FitArgs& a, const hk_fit_desc* desc) {
    return desc->model == HK_MODEL_GAIN_OFFSET && a.has_thresh && a.fail_count && !a.r2 && !a.gain && !a.flag && !a.offset_in &&
           !a.jobs && (a.use_ring == 1 || a.use_ring == 2) && (long long)desc->kh * desc->kw <= 65535;
}

static int ensure_open_rows(hk_ctx* ctx, Slot& sl, size_t words) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (sl.open_rows && sl.open_rows_words >= words) return HK_OK;
    if (sl.open_rows) {
        HK_HIP(hipStreamSynchronize(sl.stream));
        HK_HIP(dev_free(sl.open_rows));
        sl.open_rows = nullptr, sl.open_rows_words = 0;
    }
    void* p = nullptr;
    if (dev_malloc(&p, words * sizeof(unsigned)) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", words * sizeof(unsigned));
    sl.open_rows = static_cast<unsigned*>(p), sl.open_rows_words = words;
    return HK_OK;
}

// the fused launch of one job (fill_args + fill_grid done): certificate build + list launch where they apply, the one build otherwise
static int launch_fit(hk_ctx* ctx, Slot& sl, hk::FitArgs& a, const hk_fit_desc* desc, bool r2) {
    a.cert_only = 0, a.open_rows = nullptr, a.list_mode = 0;
    if (!cert_list_eligible(a, desc)) {
        HK_HIP(hk::launch_fit_apply(a, desc->model, r2, sl.stream));
        return HK_OK;
    }
    const size_t words = (size_t)a.n_bands * (size_t)a.n_strips * (size_t)((a.height + 31) / 32);
    const int rc = ensure_open_rows(ctx, sl, words);
    if (rc) return rc;
    hk::FitArgs c = a;
    c.open_rows = sl.open_rows;
    HK_HIP(hipMemsetAsync(c.open_rows, 0, words * sizeof(unsigned), sl.stream));
    c.cert_only = 1;
    HK_HIP(hk::launch_fit_apply(c, desc->model, r2, sl.stream));
    c.cert_only = 0, c.list_mode = 1;
    HK_HIP(hk::launch_fit_apply(c, desc->model, r2, sl.stream));
    return HK_OK;
}

// `n_fail`: the band's r2-mask failure count.  `pre_offset` / `pre_flag` (both or neither): offsets and source flags (r2 > thresh) & (gain > 0) & valid left by the pass
// that counted the failures (FitArgs::flag) -- the in-painting then starts right away.  `drop_params`: the parameter
// planes in `a` are scratch, the closing pass need not write them.
static int inpaint_band(Slot& sl, const hk::FitArgs& a, const hk_fit_desc* desc, bool r2, size_t plane,
                        unsigned long long n_fail, bool drop_params = false, float* pre_offset = nullptr,
                        const unsigned char* pre_flag = nullptr) {
    {
        const int rc = ensure_inpaint_scratch(sl, plane, a.height, a.stride);
        if (rc) return rc;
    }
    const hipStream_t stream = sl.stream;
    char* aux = static_cast<char*>(sl.aux);
    const float *pg = a.gain, *pr = a.r2;
    float* po = a.offset;
    const unsigned char* flags = nullptr;
    if (pre_offset && pre_flag) {
        po = pre_offset, flags = pre_flag;
    } else if (!pg || !po || !pr) {
        // parameters were not materialised by the first pass: run it again for what the in-painting reads -- the offsets
        // and the source flags, which the kernel writes itself (1 byte per pixel)
        float* scratch_off = reinterpret_cast<float*>(aux);
        hk::FitArgs b = a;
        b.gain = nullptr, b.r2 = nullptr, b.offset = scratch_off, b.corr = nullptr, b.fail_count = nullptr;
        b.flag = hk::inpaint_flag_plane(aux + plane, a.height, a.stride);
        b.cert_only = 0, b.open_rows = nullptr, b.list_mode = 0;
        HK_HIP(hk::launch_fit_apply(b, desc->model, r2, stream));
        po = scratch_off, flags = b.flag;
    }
    // The failing pixels' offsets are in-painted IN PLACE (round 5; a separate `filled` plane cost a pass-through of every source
    // pixel): sources are read only where the flag is 1, targets written only where it is 0, and the closing pass -- which reads the
    // plane at the failing pixels only -- rewrites the caller's offset plane whole when there is one.
    HK_HIP(hk::launch_inpaint_offsets(po, pg, pr, desc->r2_thresh, a.stride, a.height, a.width, aux + plane, stream, flags, n_fail));
    // closing pass: the failing pixels take the in-painted offsets and recomputed gains (kernel_model.py:370-371).  Which
    // pixels failed is in the flag plane the in-painting just used, so the build WITHOUT the R2 work runs (the R2 plane, if
    // the caller keeps one, was written by the pass that counted and is not changed by the branch)
    hk::FitArgs c = a;
    c.offset_in = po;
    c.flag_in = flags ? flags : hk::inpaint_flag_plane(aux + plane, a.height, a.stride);
    c.fail_count = nullptr;  // already counted
    c.flag = nullptr;
    c.r2 = nullptr;
    c.cert_only = 0, c.open_rows = nullptr, c.list_mode = 0;
    if (drop_params) c.gain = c.offset = nullptr;  // nobody reads them after this
    HK_HIP(hk::launch_fit_apply(c, desc->model, false, stream));
    return HK_OK;
}

// Device-side KernelModel.fit (+ apply when d_corr) of one float32 block already in HBM: block statistics for
// gain-blk-offset (or the caller's norm), the fused kernel, and the in-painting branch of gain-offset
// (kernel_model.py:361-371) when valid pixels fail the r2 mask.  d_gain / d_off / d_r2 / d_corr are nullable planes.
// A fit whose r2-mask outcome has not been looked at yet (run_host reads the counter together with the outputs: one
// stream synchronisation per block when no pixel fails, which is the usual case on well-conditioned imagery)
struct FitPending {
    hk::FitArgs a;
    bool scratch_params = false;  // a.offset / a.flag are the slot's scratch (written for the in-painting only)
    bool active = false;  // gain-offset with a threshold: fit_finish() has to look at the counter
};

// second half of fit_on_device, once the failure counter is on the host: the in-painting branch (kernel_model.py:361-371).  *requeued = the output planes were (re)written by work queued here.
int fit_finish(hk_ctx* ctx, Slot& sl, const hk_fit_desc* desc, FitPending& p, unsigned long long n_fail, bool* requeued) {
    *requeued = false;
    if (!p.active) return HK_OK;
    hk::FitArgs& a = p.a;
    const bool r2 = needs_r2(desc);
    const size_t plane = (size_t)a.stride * a.height * sizeof(float);
    ctx->expect_r2_failures.store(n_fail > 0 ? 1 : 0);
    if (n_fail > 0) {
        const int rc = inpaint_band(sl, a, desc, r2, plane, n_fail, p.scratch_params, a.flag ? a.offset : nullptr, a.flag);
        if (rc) return rc;
        *requeued = true;
    }
    return HK_OK;
}

// Device-side KernelModel.fit (+ apply when d_corr) of one float32 block already in HBM: block statistics for
// gain-blk-offset (or the caller's norm), the fused kernel, and the in-painting branch of gain-offset
// (kernel_model.py:361-371) when valid pixels fail the r2 mask.  d_gain / d_off / d_r2 / d_corr are nullable planes.
// `defer` (nullable): only queue the first pass and leave the r2-mask outcome to the caller (fit_finish); otherwise the
// stream is synchronised here to read the counter.  `win` (nullable): store window {y0, y1, x0, x1} of the kernel.
int fit_on_device(hk_ctx* ctx, Slot& sl, const hk_fit_desc* desc, const double* norm_in, float* d_src, float* d_ref,
                  int32_t height, int32_t width, int64_t stride, float* d_gain, float* d_off, float* d_r2, float* d_corr,
                  double* d_norm, unsigned long long* d_fail, void* d_norm_ws, FitPending* defer = nullptr,
                  const int* win = nullptr) {
    const bool blk = desc->model == HK_MODEL_GAIN_BLK_OFFSET;
    const bool r2 = needs_r2(desc);
    const size_t plane = (size_t)stride * height * sizeof(float);
    HK_HIP(hipMemsetAsync(d_fail, 0, sizeof(unsigned long long), sl.stream));
    if (blk) {
        if (norm_in) {
            memcpy(sl.pin<double>(Slot::PIN_NORM_IN), norm_in, 2 * sizeof(double));  // (caller memory -> pinned words)
            HK_HIP(hipMemcpyAsync(d_norm, sl.pin<double>(Slot::PIN_NORM_IN), 2 * sizeof(double), hipMemcpyHostToDevice, sl.stream));
        } else {
            hk::NormArgs na;
            na.src = d_src, na.ref = d_ref, na.height = height, na.width = width, na.stride = stride;
            na.band_stride = 0, na.n_bands = 1;
            na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
            na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
            HK_HIP(hk::launch_block_norm(na, d_norm_ws, d_norm, sl.stream));
        }
    }
    FitPending local;
    FitPending& p = defer ? *defer : local;
    hk::FitArgs& a = p.a;
    memset(&a, 0, sizeof(a));
    a.src = d_src, a.ref = d_ref, a.gain = d_gain, a.offset = d_off, a.r2 = d_r2, a.corr = d_corr;
    a.norm = blk ? d_norm : nullptr;
    a.fail_count = d_fail;
    a.height = height, a.width = width, a.stride = stride, a.band_stride = 0, a.n_bands = 1;
    fill_args(a, desc, ctx->xcd_remap);
    fill_grid(a, 0);
    // the in-painting branch needs the parameters of the whole block: the store window only narrows the other models
    if (win && !a.has_thresh) a.out_y0 = win[0], a.out_y1 = win[1], a.out_x0 = win[2], a.out_x1 = win[3];
    p.scratch_params = false;
    if (a.has_thresh && desc->model == HK_MODEL_GAIN_OFFSET && ctx->expect_r2_failures.load()) {
        // what the in-painting reads -- offsets and source flags -- into the slot's scratch right away (layout of
        // inpaint_band): no second "first pass".  With the caller's own offset plane only the flags are scratch.
        int rc = ensure_inpaint_scratch(sl, plane, height, stride);
        if (rc) return rc;
        char* aux = static_cast<char*>(sl.aux);
        a.flag = hk::inpaint_flag_plane(aux + plane, height, stride);
        if (!d_off) {
            a.offset = reinterpret_cast<float*>(aux);
            p.scratch_params = !d_gain && !d_r2;
        }
    }
    {
        const int lrc = launch_fit(ctx, sl, a, desc, r2);
        if (lrc) return lrc;
    }
    p.active = a.has_thresh != 0;
    if (defer || !p.active) return HK_OK;

    // kernel_model.py:361-371: when valid pixels fail (r2 > thresh) & (gain > 0), in-paint their offsets from the
    // passing ones and recompute their gains.  Needs the count on the host (one extra stream sync per call).
    HK_HIP(hipMemcpyAsync(sl.fail_host, d_fail, sizeof(unsigned long long), hipMemcpyDeviceToHost, sl.stream));
    HK_HIP(hipStreamSynchronize(sl.stream));
    bool requeued = false;
    return fit_finish(ctx, sl, desc, p, *sl.fail_host, &requeued);
}

// The whole host-pointer path: stage in (+ typed -> float32), (norm), fused kernel, (float32 -> typed) stage out.
// `corr_out` / `params_out` nullable; `io` nullable (float32 everywhere); `ow` nullable: window of the block that is
// written to corr_out / params_out and their row / plane strides (hk_out_window), else the whole block, packed.
int run_host(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const void* src, int64_t src_stride,
             const void* ref, int64_t ref_stride, int32_t height, int32_t width, const double* norm_in, float* params_out,
             int32_t n_param_bands, void* corr_out, double* norm_out, uint64_t* r2_fail_count, bool norm_only,
             const hk_out_window* ow = nullptr) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (!src || !ref) return fail(HK_ERR_ARG, "src/ref is NULL");
    if (height < 1 || width < 1) return fail(HK_ERR_ARG, "empty raster %d x %d", height, width);
    if (src_stride < width || ref_stride < width) return fail(HK_ERR_ARG, "row stride smaller than width");
    const int sdt = io ? io->src_dtype : 0, rdt = io ? io->ref_dtype : 0, odt = io ? io->out_dtype : 0;
    if (!hk::dtype_size(sdt) || !hk::dtype_size(rdt) || !hk::dtype_size(odt)) return fail(HK_ERR_ARG, "unknown dtype");
    const bool out_cast = io && (odt != 0 || io->out_has_nodata);
    const bool r2 = needs_r2(desc);
    if (!norm_only) {
        if (params_out && n_param_bands != (r2 ? 3 : 2))
            return fail(HK_ERR_ARG, "n_param_bands must be %d for this model configuration", r2 ? 3 : 2);
        if (!params_out && !corr_out) return fail(HK_ERR_ARG, "nothing to compute: params_out and corr_out are NULL");
    }
    // output window (defaults: the whole block, written packed)
    int wr0 = 0, wc0 = 0, wrows = height, wcols = width;
    int64_t out_stride = width, par_stride = width, out_band_stride = (int64_t)height * width;
    if (ow) {
        wr0 = ow->row0, wc0 = ow->col0, wrows = ow->rows, wcols = ow->cols;
        out_stride = ow->stride, out_band_stride = ow->band_stride;
        par_stride = ow->param_stride > 0 ? ow->param_stride : ow->stride;
        if (wr0 < 0 || wc0 < 0 || wrows < 1 || wcols < 1 || wr0 + wrows > height || wc0 + wcols > width)
            return fail(HK_ERR_ARG, "output window outside the block");
        if ((corr_out && out_stride < wcols) || (params_out && par_stride < wcols))
            return fail(HK_ERR_ARG, "output row stride smaller than the window");
        if (params_out && n_param_bands > 1 && out_band_stride < (int64_t)(wrows - 1) * par_stride + wcols)
            return fail(HK_ERR_ARG, "output band stride smaller than a plane of the window");
    }
    HK_ENTER(ctx);

    const int64_t stride = (width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;
    const size_t plane = (size_t)stride * height * sizeof(float);
    const bool blk = desc->model == HK_MODEL_GAIN_BLK_OFFSET;
    const bool want_norm = blk || norm_only;

    // bump allocation inside the stream's device slab (everything 256-byte aligned)
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t off = total; total += (bytes + 255) / 256 * 256; return off; };
    const size_t o_src = take(plane), o_ref = take(plane);
    size_t o_gain = 0, o_off = 0, o_r2 = 0, o_corr = 0, o_raw_s = 0, o_raw_r = 0, o_raw_o = 0;
    if (!norm_only) {
        if (params_out) {
            o_gain = take(plane), o_off = take(plane);
            if (n_param_bands == 3) o_r2 = take(plane);
        }
        if (corr_out) o_corr = take(plane);
        if (corr_out && out_cast) o_raw_o = take((size_t)stride * height * hk::dtype_size(odt));
    }
    if (sdt) o_raw_s = take((size_t)stride * height * hk::dtype_size(sdt));
    if (rdt) o_raw_r = take((size_t)stride * height * hk::dtype_size(rdt));
    const size_t o_aux = take(256);
    const size_t o_ws = want_norm ? take(hk::norm_workspace_bytes(1, height, width)) : 0;

    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    rc = ensure_dev(sl, total);
    if (rc) return rc;
    char* base = static_cast<char*>(sl.dev);
    float* d_src = reinterpret_cast<float*>(base + o_src);
    float* d_ref = reinterpret_cast<float*>(base + o_ref);
    float* d_gain = (!norm_only && params_out) ? reinterpret_cast<float*>(base + o_gain) : nullptr;
    float* d_off = (!norm_only && params_out) ? reinterpret_cast<float*>(base + o_off) : nullptr;
    float* d_r2 = (!norm_only && params_out && n_param_bands == 3) ? reinterpret_cast<float*>(base + o_r2) : nullptr;
    float* d_corr = (!norm_only && corr_out) ? reinterpret_cast<float*>(base + o_corr) : nullptr;
    double* d_norm = reinterpret_cast<double*>(base + o_aux);  // 2 doubles
    unsigned long long* d_fail = reinterpret_cast<unsigned long long*>(base + o_aux + 64);
    void* d_ws = base + o_ws;

    // stage in: float32 planes directly, other dtypes through a raw plane + on-device conversion
    auto stage_in = [&](const void* host, int64_t hstride, int dt, float* dplane, size_t o_raw) -> int {
        const size_t es = hk::dtype_size(dt);
        void* dst = dt ? static_cast<void*>(base + o_raw) : static_cast<void*>(dplane);
        const int src_rc = stage_h2d(sl, dst, stride * es, host, hstride * es, (size_t)width * es, height);
        if (src_rc) return src_rc;
        if (dt) HK_HIP(hk::launch_cast_in(dt, dst, stride, dplane, stride, height, width, sl.stream));
        return HK_OK;
    };
    rc = stage_in(src, src_stride, sdt, d_src, o_raw_s);
    if (rc) return rc;
    rc = stage_in(ref, ref_stride, rdt, d_ref, o_raw_r);
    if (rc) return rc;
    if (norm_only) {
        hk::NormArgs na;
        na.src = d_src, na.ref = d_ref, na.height = height, na.width = width, na.stride = stride;
        na.band_stride = 0, na.n_bands = 1;
        na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
        na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
        HK_HIP(hk::launch_block_norm(na, d_ws, d_norm, sl.stream));
        HK_HIP(hipMemcpyAsync(sl.pin<double>(Slot::PIN_NORM), d_norm, 2 * sizeof(double), hipMemcpyDeviceToHost, sl.stream));
        if ((rc = stage_finish(sl))) return rc;
        if (norm_out) memcpy(norm_out, sl.pin<double>(Slot::PIN_NORM), 2 * sizeof(double));
        return HK_OK;
    }

    // stage out: the window of the block the caller wants, straight into its (strided) arrays
    const size_t win_off = (size_t)wr0 * stride + wc0;  // elements
    auto stage_out = [&]() -> int {
        const size_t wbytes = (size_t)wcols * sizeof(float);
        float* outs[3] = {d_gain, d_off, d_r2};
        if (params_out)
            for (int b = 0; b < n_param_bands; ++b) {
                const int prc = stage_d2h(sl, params_out + (size_t)b * out_band_stride, par_stride * sizeof(float),
                                          outs[b] + win_off, stride * sizeof(float), wbytes, wrows);
                if (prc) return prc;
            }
        if (corr_out) {
            if (out_cast) {
                const size_t es = hk::dtype_size(odt);
                char* d_raw = base + o_raw_o;
                HK_HIP(hk::launch_cast_out(odt, d_corr, stride, d_raw, stride, height, width, io->out_has_nodata,
                                           io->out_nodata, sl.stream));
                return stage_d2h(sl, corr_out, out_stride * es, d_raw + win_off * es, stride * es, (size_t)wcols * es, wrows);
            }
            return stage_d2h(sl, corr_out, out_stride * sizeof(float), d_corr + win_off, stride * sizeof(float), wbytes, wrows);
        }
        return HK_OK;
    };

    // The fit, its outputs and its r2-mask counter are queued back to back and the stream is synchronised ONCE; only
    // when pixels failed the mask (or the certificate-only build asks for its re-run) do the in-painting passes follow,
    // and the outputs are copied again behind them.
    FitPending pending;
    const int kwin[4] = {wr0, wr0 + wrows, wc0 / hk::PX * hk::PX, width};  // rows exactly, columns from the window's first quad
    rc = fit_on_device(ctx, sl, desc, norm_in, d_src, d_ref, height, width, stride, d_gain, d_off, d_r2, d_corr, d_norm,
                       d_fail, d_ws, &pending, (ow && !out_cast) ? kwin : nullptr);
    if (rc) return rc;
    if (norm_out && blk)
        HK_HIP(hipMemcpyAsync(sl.pin<double>(Slot::PIN_NORM), d_norm, 2 * sizeof(double), hipMemcpyDeviceToHost, sl.stream));
    if ((rc = stage_out())) return rc;
    // hk_debug_fail_after_d2h(1) (homonim_hk_devtools.h): the call fails HERE, its result copies queued and not yet unpacked --
    // the state every HIP error between a stage_d2h and stage_finish leaves (tests/test_gpu_staging.py)
    if (g_fail_after_d2h.load(std::memory_order_relaxed))
        return fail(HK_ERR_HIP, "hk_debug_fail_after_d2h: injected failure behind the result copies");
    *sl.fail_host = 0;
    if (pending.active) HK_HIP(hipMemcpyAsync(sl.fail_host, d_fail, sizeof(unsigned long long), hipMemcpyDeviceToHost, sl.stream));
    if ((rc = stage_finish(sl))) return rc;
    if (norm_out && blk) memcpy(norm_out, sl.pin<double>(Slot::PIN_NORM), 2 * sizeof(double));
    unsigned long long n_fail = *sl.fail_host;
    if (pending.active) {
        bool requeued = false;
        rc = fit_finish(ctx, sl, desc, pending, n_fail, &requeued);
        if (rc) return rc;
        if (requeued) {
            if ((rc = stage_out())) return rc;
            HK_HIP(hipMemcpyAsync(sl.fail_host, d_fail, sizeof(unsigned long long), hipMemcpyDeviceToHost, sl.stream));
            if ((rc = stage_finish(sl))) return rc;
            n_fail = *sl.fail_host;  // (the in-painting passes do not count; a re-run of the complete build does)
        }
    }
    if (r2_fail_count) *r2_fail_count = n_fail;
    return HK_OK;
}

}  // namespace

extern "C" {

int hk_abi_version(void) { return HK_ABI_VERSION; }
const char* hk_backend_name(void) { return "hip-gfx950"; }
const char* hk_last_error(void) { return g_err; }

int hk_device_count(int* count) {
    if (!count) return fail(HK_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(HK_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return HK_OK;
}

int hk_device_pci_bus_id(int device_id, char* out, int len) {
    if (!out || len < 13) return fail(HK_ERR_ARG, "bus-id buffer is NULL or shorter than 13 bytes");
    out[0] = '\0';
    hipError_t e = hipDeviceGetPCIBusId(out, len, device_id);
    if (e != hipSuccess) return fail(HK_ERR_NODEVICE, "hipDeviceGetPCIBusId(%d): %s", device_id, hipGetErrorString(e));
    for (char* c = out; *c; ++c) *c = (char)tolower((unsigned char)*c);   // sysfs spells bus addresses in lower case
    return HK_OK;
}

int hk_ctx_create(int device_id, int n_streams, hk_ctx** out) {
    if (!out) return fail(HK_ERR_ARG, "ctx out-pointer is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
        return fail(HK_ERR_NODEVICE, "no HIP device available (libhomonim_hk needs an MI355X / gfx950 GPU)");
    if (device_id < 0 || device_id >= n) return fail(HK_ERR_ARG, "device %d out of range [0, %d)", device_id, n);
    if (n_streams < 1) n_streams = 1;
    if (n_streams > 64) n_streams = 64;
    HK_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HK_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    hk_ctx* ctx = new (std::nothrow) hk_ctx();
    if (!ctx) return fail(HK_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    ctx->slots.resize(n_streams);
    const char* remap = getenv("HK_XCD_REMAP");
    // runs of this many consecutive units (neighbouring strips) per XCD, 0 = plain round-robin (hk_fit_kernel.h); 16 measured
    // best across models on MI355X (gain 5x5: -12 %, gain-offset without the r2 mask: -4 %, VALU-bound variants: -1 %)
    ctx->xcd_remap = remap ? std::min(std::max(atoi(remap), 0), 256) : 16;
    for (auto& s : ctx->slots) {
        hipError_t e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            hk_ctx_destroy(ctx);
            return fail(HK_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        if (ensure_pin(s) != HK_OK) {
            hk_ctx_destroy(ctx);
            return HK_ERR_NOMEM;
        }
    }
    {
        hipError_t e = hipStreamCreateWithFlags(&ctx->xfer.stream, hipStreamNonBlocking);
        if (e != hipSuccess || ensure_pin(ctx->xfer) != HK_OK) {
            hk_ctx_destroy(ctx);
            return e != hipSuccess ? fail(HK_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)) : HK_ERR_NOMEM;
        }
    }
    register_gpu_fault_report();  // (HIP, and with it the HSA runtime, is up: the streams above were created on it)
    *out = ctx;
    return HK_OK;
}

int hk_ctx_destroy(hk_ctx* ctx) {
    if (!ctx) return HK_OK;
    (void)hipSetDevice(ctx->device);
    for (auto& s : ctx->slots) slot_release(s);
    slot_release(ctx->xfer);
    if (ctx->comm && rccl().ok) rccl().CommDestroy(ctx->comm);
    delete ctx;
    return HK_OK;
}

int hk_ctx_sync(hk_ctx* ctx) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    HK_ENTER(ctx);
    HK_HIP(hipDeviceSynchronize());
    return HK_OK;
}

int hk_block_norm(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
                  int64_t ref_stride, int32_t height, int32_t width, double norm_out[2]) {
    if (!norm_out) return fail(HK_ERR_ARG, "norm_out is NULL");
    return run_host(ctx, desc, nullptr, src, src_stride, ref, ref_stride, height, width, nullptr, nullptr, 0, nullptr,
                    norm_out, nullptr, true);
}

int hk_fit(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
           int64_t ref_stride, int32_t height, int32_t width, const double* norm_in, float* params_out,
           int32_t n_param_bands, double* norm_out, uint64_t* r2_fail_count) {
    if (!params_out) return fail(HK_ERR_ARG, "params_out is NULL");
    return run_host(ctx, desc, nullptr, src, src_stride, ref, ref_stride, height, width, norm_in, params_out,
                    n_param_bands, nullptr, norm_out, r2_fail_count, false);
}

int hk_fit_apply(hk_ctx* ctx, const hk_fit_desc* desc, const float* src, int64_t src_stride, const float* ref,
                 int64_t ref_stride, int32_t height, int32_t width, const double* norm_in, float* params_out,
                 int32_t n_param_bands, float* corr_out, double* norm_out, uint64_t* r2_fail_count) {
    if (!corr_out) return fail(HK_ERR_ARG, "corr_out is NULL");
    return run_host(ctx, desc, nullptr, src, src_stride, ref, ref_stride, height, width, norm_in, params_out,
                    n_param_bands, corr_out, norm_out, r2_fail_count, false);
}

int hk_fit_apply_io(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const void* src, int64_t src_stride,
                    const void* ref, int64_t ref_stride, int32_t height, int32_t width, const double* norm_in,
                    float* params_out, int32_t n_param_bands, void* corr_out, double* norm_out, uint64_t* r2_fail_count) {
    return run_host(ctx, desc, io, src, src_stride, ref, ref_stride, height, width, norm_in, params_out, n_param_bands,
                    corr_out, norm_out, r2_fail_count, false);
}

int hk_fit_apply_block(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const void* src, int64_t src_stride,
                       const void* ref, int64_t ref_stride, int32_t height, int32_t width, const double* norm_in,
                       float* params_out, int32_t n_param_bands, void* corr_out, const hk_out_window* window,
                       double* norm_out, uint64_t* r2_fail_count) {
    return run_host(ctx, desc, io, src, src_stride, ref, ref_stride, height, width, norm_in, params_out, n_param_bands,
                    corr_out, norm_out, r2_fail_count, false, window);
}

int hk_apply(hk_ctx* ctx, const float* src, int64_t src_stride, const float* params, int32_t height, int32_t width,
             float* out) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!src || !params || !out) return fail(HK_ERR_ARG, "NULL pointer argument");
    if (height < 1 || width < 1) return fail(HK_ERR_ARG, "empty raster %d x %d", height, width);
    if (src_stride < width) return fail(HK_ERR_ARG, "row stride smaller than width");
    HK_ENTER(ctx);
    const int64_t stride = (width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;
    const size_t plane = (size_t)stride * height * sizeof(float);
    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    int rc = ensure_dev(sl, 4 * plane);
    if (rc) return rc;
    char* base = static_cast<char*>(sl.dev);
    float* d_src = reinterpret_cast<float*>(base);
    float* d_gain = reinterpret_cast<float*>(base + plane);
    float* d_off = reinterpret_cast<float*>(base + 2 * plane);
    float* d_out = reinterpret_cast<float*>(base + 3 * plane);
    const size_t wbytes = (size_t)width * sizeof(float);
    if ((rc = stage_h2d(sl, d_src, stride * 4, src, src_stride * 4, wbytes, height))) return rc;
    if ((rc = stage_h2d(sl, d_gain, stride * 4, params, wbytes, wbytes, height))) return rc;
    if ((rc = stage_h2d(sl, d_off, stride * 4, params + (size_t)height * width, wbytes, wbytes, height))) return rc;
    HK_HIP(hk::launch_apply(d_src, d_gain, d_off, d_out, height, width, stride, sl.stream));
    if ((rc = stage_d2h(sl, out, wbytes, d_out, stride * 4, wbytes, height))) return rc;
    return stage_finish(sl);
}

int hk_refspace_fit_apply(hk_ctx* ctx, const hk_fit_desc* desc, const hk_io_desc* io, const hk_space_desc* space,
                          const void* src, int64_t src_stride, int32_t src_height, int32_t src_width, const void* ref,
                          int64_t ref_stride, int32_t ref_height, int32_t ref_width, float* params_out,
                          int32_t n_param_bands, void* corr_out, uint64_t* r2_fail_count) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (!space || !src || !ref || !corr_out) return fail(HK_ERR_ARG, "NULL pointer argument");
    if (src_height < 1 || src_width < 1 || ref_height < 1 || ref_width < 1) return fail(HK_ERR_ARG, "empty raster");
    if (src_stride < src_width || ref_stride < ref_width) return fail(HK_ERR_ARG, "row stride smaller than width");
    for (int m : {space->down_resampling, space->up_resampling})
        if (!resampling_built(m)) return fail(HK_ERR_UNSUPPORTED, "resampling %d is not built", m);
    if (!(space->down[0] > 0 && space->down[2] > 0 && space->up[0] > 0 && space->up[2] > 0))
        return fail(HK_ERR_UNSUPPORTED, "flipped or degenerate grid mapping");
    if (src_height > 65535 || ref_height > 65535) return fail(HK_ERR_UNSUPPORTED, "block taller than 65535 rows");
    const int sdt = io ? io->src_dtype : 0, rdt = io ? io->ref_dtype : 0, odt = io ? io->out_dtype : 0;
    if (!hk::dtype_size(sdt) || !hk::dtype_size(rdt) || !hk::dtype_size(odt)) return fail(HK_ERR_ARG, "unknown dtype");
    const bool out_cast = io && (odt != 0 || io->out_has_nodata);
    const bool r2 = needs_r2(desc);
    if (params_out && n_param_bands != (r2 ? 3 : 2))
        return fail(HK_ERR_ARG, "n_param_bands must be %d for this model configuration", r2 ? 3 : 2);
    HK_ENTER(ctx);

    const int64_t ss = (src_width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;  // source-grid row stride
    const int64_t rs = (ref_width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;  // reference-grid row stride
    const size_t splane = (size_t)ss * src_height * 4, rplane = (size_t)rs * ref_height * 4;
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t off = total; total += (bytes + 255) / 256 * 256; return off; };
    const size_t o_src = take(splane), o_ref = take(rplane), o_ds = take(rplane);
    const size_t o_gain = take(rplane), o_off = take(rplane), o_r2 = r2 ? take(rplane) : 0;
    // bilinear / cubic_spline parameters are up-sampled inside the apply kernel (no full-resolution parameter planes)
    const bool fused_up = (space->up_resampling == 1 || space->up_resampling == 3) && ref_width >= 4 &&
                          space->up[0] <= 1.0 + 1e-9 && space->up[2] <= 1.0 + 1e-9 &&
                          (long long)ref_height * rs < 0x7fffffffLL;
    const size_t o_gus = fused_up ? 0 : take(splane), o_ous = fused_up ? 0 : take(splane), o_corr = take(splane);
    const size_t o_rowtab = fused_up ? take(hk::upsample_apply_workspace_bytes(src_height)) : 0;
    const size_t o_vs = space->mask_partial ? take(splane) : 0, o_cov = space->mask_partial ? take(rplane) : 0;
    const size_t o_mk = space->mask_partial ? take((size_t)rs * ref_height) : 0;
    const size_t o_mkf = space->mask_partial ? take(rplane) : 0, o_keep = space->mask_partial ? take(splane) : 0;
    const size_t o_cnt = space->mask_partial ? take((size_t)rs * ref_height * 2) : 0;
    const size_t o_raw_s = sdt ? take((size_t)ss * src_height * hk::dtype_size(sdt)) : 0;
    const size_t o_raw_r = rdt ? take((size_t)rs * ref_height * hk::dtype_size(rdt)) : 0;
    const size_t o_raw_o = out_cast ? take((size_t)ss * src_height * hk::dtype_size(odt)) : 0;
    const size_t o_aux = take(256);
    const bool blk = desc->model == HK_MODEL_GAIN_BLK_OFFSET;
    const size_t o_ws = blk ? take(hk::norm_workspace_bytes(1, ref_height, ref_width)) : 0;

    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    rc = ensure_dev(sl, total);
    if (rc) return rc;
    char* base = static_cast<char*>(sl.dev);
    auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    float *d_src = F(o_src), *d_ref = F(o_ref), *d_ds = F(o_ds), *d_gain = F(o_gain), *d_off = F(o_off);
    float *d_r2 = r2 ? F(o_r2) : nullptr, *d_gus = F(o_gus), *d_ous = F(o_ous), *d_corr = F(o_corr);
    double* d_norm = reinterpret_cast<double*>(base + o_aux);
    unsigned long long* d_fail = reinterpret_cast<unsigned long long*>(base + o_aux + 64);
    const float nan = std::nanf("");

    auto stage_in = [&](const void* host, int64_t hstride, int dt, float* dplane, size_t o_raw, int64_t dstride, int h,
                        int w) -> int {
        const size_t es = hk::dtype_size(dt);
        void* dst = dt ? static_cast<void*>(base + o_raw) : static_cast<void*>(dplane);
        const int src_rc = stage_h2d(sl, dst, dstride * es, host, hstride * es, (size_t)w * es, h);
        if (src_rc) return src_rc;
        if (dt) HK_HIP(hk::launch_cast_in(dt, dst, dstride, dplane, dstride, h, w, sl.stream));
        return HK_OK;
    };
    if ((rc = stage_in(src, src_stride, sdt, d_src, o_raw_s, ss, src_height, src_width))) return rc;
    if ((rc = stage_in(ref, ref_stride, rdt, d_ref, o_raw_r, rs, ref_height, ref_width))) return rc;

    // RefSpaceModel.fit (:476-482): source -> reference grid (nodata nan), then the base-class fit there
    HK_HIP(hk::launch_resample(space->down_resampling, d_src, ss, 0, src_height, src_width, 1, desc->src_nodata_mode,
                               desc->src_nodata, space->down[0], space->down[1], space->down[2], space->down[3], d_ds, rs, 0,
                               ref_height, ref_width, nan, sl.stream));
    hk_fit_desc fd = *desc;
    fd.src_nodata_mode = HK_NODATA_NAN, fd.src_nodata = nan;
    rc = fit_on_device(ctx, sl, &fd, nullptr, d_ds, d_ref, ref_height, ref_width, rs, d_gain, d_off, d_r2, nullptr, d_norm,
                       d_fail, base + o_ws);
    if (rc) return rc;

    // RefSpaceModel.apply (:484-503): gain / offset -> source grid, re-mask, apply
    if (!fused_up)
        for (int b = 0; b < 2; ++b)
            HK_HIP(hk::launch_resample(space->up_resampling, b ? d_off : d_gain, rs, 0, ref_height, ref_width, 1,
                                       HK_NODATA_NAN, nan, space->up[0], space->up[1], space->up[2], space->up[3],
                                       b ? d_ous : d_gus, ss, 0, src_height, src_width, nan, sl.stream));
    const float* d_keep = nullptr;
    if (space->mask_partial) {
        // _full_coverage_mask (:375-409): source mask --average--> reference grid (>= 1) & parameter mask, eroded by
        // (kh+2) x (kw+2); back to the source grid with `nearest` (nodata 0)
        HK_HIP(hk::launch_valid_plane(d_src, ss, desc->src_nodata_mode, desc->src_nodata, F(o_vs), ss, src_height,
                                      src_width, sl.stream));
        HK_HIP(hk::launch_resample(5, F(o_vs), ss, 0, src_height, src_width, 1, HK_NODATA_NONE, 0.f, space->down[0],
                                   space->down[1], space->down[2], space->down[3], F(o_cov), rs, 0, ref_height, ref_width,
                                   0.f, sl.stream));
        // parameters sit in consecutive planes only by construction of the bump allocator above (gain, offset)
        if (o_off != o_gain + ((rplane + 255) / 256 * 256)) return fail(HK_ERR_HIP, "internal: parameter planes not adjacent");
        unsigned char* d_mk = reinterpret_cast<unsigned char*>(base + o_mk);
        HK_HIP(hk::launch_partial_mask(F(o_cov), 3, 0.f, d_gain, 2, (long long)((rplane + 255) / 256 * 256 / 4), nullptr,
                                       ref_height, ref_width, rs, desc->kh, desc->kw,
                                       reinterpret_cast<unsigned short*>(base + o_cnt), nullptr, nullptr, d_mk, sl.stream));
        HK_HIP(hk::launch_cast_in(1, d_mk, rs, F(o_mkf), rs, ref_height, ref_width, sl.stream));
        HK_HIP(hk::launch_resample(0, F(o_mkf), rs, 0, ref_height, ref_width, 1, HK_NODATA_NONE, 0.f, space->up[0],
                                   space->up[1], space->up[2], space->up[3], F(o_keep), ss, 0, src_height, src_width, 0.f,
                                   sl.stream));
        d_keep = F(o_keep);
    }
    if (fused_up) {
        HK_HIP(hk::launch_upsample_apply(space->up_resampling, d_src, ss, desc->src_nodata_mode, desc->src_nodata, d_gain,
                                         d_off, rs, ref_height, ref_width, d_keep, ss, d_corr, ss, src_height, src_width,
                                         space->up[0], space->up[1], space->up[2], space->up[3], base + o_rowtab, sl.stream));
    } else {
        HK_HIP(hk::launch_apply_space(d_src, ss, desc->src_nodata_mode, desc->src_nodata, d_gus, d_ous, ss, d_keep, d_corr,
                                      ss, src_height, src_width, sl.stream));
    }

    if (params_out) {
        float* outs[3] = {d_gain, d_off, d_r2};
        const size_t wb = (size_t)ref_width * 4;
        for (int b = 0; b < n_param_bands; ++b)
            if ((rc = stage_d2h(sl, params_out + (size_t)b * ref_height * ref_width, wb, outs[b], rs * 4, wb, ref_height))) return rc;
    }
    if (out_cast) {
        const size_t es = hk::dtype_size(odt);
        void* d_raw = base + o_raw_o;
        HK_HIP(hk::launch_cast_out(odt, d_corr, ss, d_raw, ss, src_height, src_width, io->out_has_nodata, io->out_nodata,
                                   sl.stream));
        if ((rc = stage_d2h(sl, corr_out, (size_t)src_width * es, d_raw, ss * es, (size_t)src_width * es, src_height))) return rc;
    } else {
        if ((rc = stage_d2h(sl, corr_out, (size_t)src_width * 4, d_corr, ss * 4, (size_t)src_width * 4, src_height))) return rc;
    }
    HK_HIP(hipMemcpyAsync(sl.fail_host, d_fail, sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream));
    if ((rc = stage_finish(sl))) return rc;
    if (r2_fail_count) *r2_fail_count = *sl.fail_host;
    return HK_OK;
}

int hk_reproject(hk_ctx* ctx, const float* src, int32_t n_bands, int32_t src_height, int32_t src_width,
                 int32_t src_nodata_mode, float src_nodata, double kx, double ox, double ky, double oy, int32_t resampling,
                 float* dst, int32_t dst_height, int32_t dst_width, float dst_fill) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!src || !dst) return fail(HK_ERR_ARG, "NULL pointer argument");
    if (n_bands < 1 || src_height < 1 || src_width < 1 || dst_height < 1 || dst_width < 1)
        return fail(HK_ERR_ARG, "empty raster");
    if (!(kx > 0.0) || !(ky > 0.0)) return fail(HK_ERR_UNSUPPORTED, "flipped or degenerate grid mapping");
    if (!resampling_built(resampling))
        return fail(HK_ERR_UNSUPPORTED, "resampling %d is not a warp method (GRA_* codes 0..6 and 8..14 are built; 7 = gauss is "
                                        "an overview-only method in GDAL / rasterio as well)", resampling);
    if (dst_height > 65535) return fail(HK_ERR_UNSUPPORTED, "destination taller than 65535 rows");
    HK_ENTER(ctx);
    const size_t sbytes = (size_t)n_bands * src_height * src_width * 4, dbytes = (size_t)n_bands * dst_height * dst_width * 4;
    const size_t o_dst = (sbytes + 255) / 256 * 256;
    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    int rc = ensure_dev(sl, o_dst + dbytes);
    if (rc) return rc;
    float* d_src = static_cast<float*>(sl.dev);
    float* d_dst = reinterpret_cast<float*>(static_cast<char*>(sl.dev) + o_dst);
    if ((rc = stage_h2d(sl, d_src, sbytes, src, sbytes, sbytes, 1))) return rc;
    HK_HIP(hk::launch_resample(resampling, d_src, src_width, (long long)src_height * src_width, src_height, src_width,
                               n_bands, src_nodata_mode, src_nodata, kx, ox, ky, oy, d_dst, dst_width,
                               (long long)dst_height * dst_width, dst_height, dst_width, dst_fill, sl.stream));
    if ((rc = stage_d2h(sl, dst, dbytes, d_dst, dbytes, dbytes, 1))) return rc;
    return stage_finish(sl);
}

int hk_partial_mask(hk_ctx* ctx, const float* in, int64_t in_stride, int32_t in_nodata_mode, float in_nodata,
                    const float* params, int32_t n_param_bands, const float* src, int64_t src_stride, int32_t height,
                    int32_t width, int32_t kh, int32_t kw, float* params_out, float* corr_out, uint8_t* mask_out) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!in || !params) return fail(HK_ERR_ARG, "NULL pointer argument");
    if (height < 1 || width < 1) return fail(HK_ERR_ARG, "empty raster %d x %d", height, width);
    if (n_param_bands < 2 || n_param_bands > 3) return fail(HK_ERR_ARG, "params must have 2 or 3 bands");
    if (kh < 1 || kw < 1 || !(kh & 1) || !(kw & 1)) return fail(HK_ERR_ARG, "`kernel_shape` must be odd in both dimensions.");
    if ((kh + 2) * (kw + 2) > 65535) return fail(HK_ERR_UNSUPPORTED, "kernel too large for mask_partial");
    if (corr_out && !src) return fail(HK_ERR_ARG, "corr_out needs src");
    if (in_stride < width || (src && src_stride < width)) return fail(HK_ERR_ARG, "row stride smaller than width");
    HK_ENTER(ctx);
    const int64_t stride = (width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;
    const size_t plane = (size_t)stride * height * sizeof(float);
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t off = total; total += (bytes + 255) / 256 * 256; return off; };
    const size_t o_in = take(plane), o_par = take(plane * n_param_bands), o_src = src ? take(plane) : 0;
    const size_t o_pout = params_out ? take(plane * n_param_bands) : 0, o_corr = corr_out ? take(plane) : 0;
    const size_t o_mask = mask_out ? take((size_t)stride * height) : 0, o_cnt = take((size_t)stride * height * 2);
    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    int rc = ensure_dev(sl, total);
    if (rc) return rc;
    char* base = static_cast<char*>(sl.dev);
    float* d_in = reinterpret_cast<float*>(base + o_in);
    float* d_par = reinterpret_cast<float*>(base + o_par);
    float* d_src = src ? reinterpret_cast<float*>(base + o_src) : nullptr;
    float* d_pout = params_out ? reinterpret_cast<float*>(base + o_pout) : nullptr;
    float* d_corr = corr_out ? reinterpret_cast<float*>(base + o_corr) : nullptr;
    unsigned char* d_mask = mask_out ? reinterpret_cast<unsigned char*>(base + o_mask) : nullptr;
    const size_t wb = (size_t)width * 4, sb = (size_t)stride * 4;
    if ((rc = stage_h2d(sl, d_in, sb, in, in_stride * 4, wb, height))) return rc;
    for (int b = 0; b < n_param_bands; ++b)
        if ((rc = stage_h2d(sl, d_par + (size_t)b * stride * height, sb, params + (size_t)b * height * width, wb, wb, height)))
            return rc;
    if (src && (rc = stage_h2d(sl, d_src, sb, src, src_stride * 4, wb, height))) return rc;
    HK_HIP(hk::launch_partial_mask(d_in, in_nodata_mode, in_nodata, d_par, n_param_bands, stride * height, d_src, height,
                                   width, stride, kh, kw, reinterpret_cast<unsigned short*>(base + o_cnt), d_pout, d_corr,
                                   d_mask, sl.stream));
    if (params_out)
        for (int b = 0; b < n_param_bands; ++b)
            if ((rc = stage_d2h(sl, params_out + (size_t)b * height * width, wb, d_pout + (size_t)b * stride * height, sb, wb, height)))
                return rc;
    if (corr_out && (rc = stage_d2h(sl, corr_out, wb, d_corr, sb, wb, height))) return rc;
    if (mask_out && (rc = stage_d2h(sl, mask_out, width, d_mask, stride, width, height))) return rc;
    return stage_finish(sl);
}

// ---------------------------------------------------------------------------------------------------------------------
int hk_host_alloc(hk_ctx* ctx, size_t bytes, void** hptr) {
    if (!ctx || !hptr) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    if (hipHostMalloc(hptr, bytes, hipHostMallocPortable) != hipSuccess)
        return fail(HK_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
    pinned_ranges().add(*hptr, bytes);
    return HK_OK;
}
int hk_host_free(hk_ctx* ctx, void* hptr) {
    // page-locked host memory belongs to no device: `ctx` may be NULL (e.g. the context was destroyed first)
    (void)ctx;
    pinned_ranges().remove(hptr);
    HK_HIP(hipHostFree(hptr));
    return HK_OK;
}
int hk_host_register(hk_ctx* ctx, void* hptr, size_t bytes) {
    if (!ctx || !hptr) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    // page-locked already (hk_host_alloc, or somebody's registration)?  hipHostRegister answers that case with different
    // errors (AlreadyRegistered, InvalidValue for hipHostMalloc memory) or not at all, so ask first
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, hptr) == hipSuccess && attr.type == hipMemoryTypeHost)
        return fail(HK_ERR_ALREADY, "host memory is page-locked already");
    (void)hipGetLastError();  // pageable memory: the query itself may have failed
    hipError_t e = hipHostRegister(hptr, bytes, hipHostRegisterPortable);
    if (e == hipErrorHostMemoryAlreadyRegistered) {
        (void)hipGetLastError();
        return fail(HK_ERR_ALREADY, "host memory is page-locked already");
    }
    HK_HIP(e);
    pinned_ranges().add(hptr, bytes);
    return HK_OK;
}
int hk_host_unregister(hk_ctx* ctx, void* hptr) {
    if (!hptr) return fail(HK_ERR_ARG, "NULL argument");
    (void)ctx;  // may be NULL, like hk_host_free
    pinned_ranges().remove(hptr);
    HK_HIP(hipHostUnregister(hptr));
    return HK_OK;
}
int hk_debug_staging_counters(uint64_t out[2], int32_t reset) {
    if (!out) return fail(HK_ERR_ARG, "out is NULL");
    out[0] = g_direct_copies.load(), out[1] = g_staged_chunks.load();
    if (reset) g_direct_copies.store(0), g_staged_chunks.store(0);
    return HK_OK;
}

int hk_debug_fail_after_d2h(int32_t on) {
    g_fail_after_d2h.store(on != 0);
    return HK_OK;
}

int hk_debug_build_ledger(char* buf, size_t len, size_t* needed, int32_t reset) {
    const size_t n = hk::ledger_text(buf, buf ? len : 0, reset != 0);
    if (needed) *needed = n;
    if (buf && n > len) return fail(HK_ERR_ARG, "ledger needs %zu bytes, the buffer has %zu", n, len);
    return HK_OK;
}

int hk_dev_alloc(hk_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    if (dev_malloc(dptr, bytes) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
    return HK_OK;
}
int hk_dev_free(hk_ctx* ctx, void* dptr) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    HK_ENTER(ctx);
    HK_HIP(dev_free(dptr));
    return HK_OK;
}
int hk_memcpy_h2d(hk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!bytes) return HK_OK;
    if (!dst || !src) return fail(HK_ERR_ARG, "NULL pointer argument");
    HK_ENTER(ctx);
    std::lock_guard<std::mutex> lk(ctx->xfer_mu);
    int rc = stage_h2d(ctx->xfer, dst, bytes, src, bytes, bytes, 1);
    if (!rc) rc = stage_finish(ctx->xfer);
    if (rc) stage_abandon(ctx->xfer);
    return rc;
}
int hk_memcpy_d2h(hk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!bytes) return HK_OK;
    if (!dst || !src) return fail(HK_ERR_ARG, "NULL pointer argument");
    HK_ENTER(ctx);
    std::lock_guard<std::mutex> lk(ctx->xfer_mu);
    int rc = stage_d2h(ctx->xfer, dst, bytes, src, bytes, bytes, 1);
    if (!rc) rc = stage_finish(ctx->xfer);
    if (rc) stage_abandon(ctx->xfer);   // (a second chunk that failed leaves the first one queued with the caller's pointer)
    return rc;
}
int hk_memset(hk_ctx* ctx, void* dst, int value, size_t bytes) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    HK_ENTER(ctx);
    HK_HIP(hipMemset(dst, value, bytes));
    return HK_OK;
}

// elements spanned by the job's planes (last band's plane included)
static size_t job_span(int32_t n_bands, int32_t height, int64_t stride, int64_t band_stride) {
    const size_t plane = (size_t)stride * (size_t)height;
    return n_bands > 1 ? (size_t)(n_bands - 1) * (size_t)band_stride + plane : plane;
}
uint64_t hk_dev_job_scratch_bytes(int32_t n_bands, int32_t height, int64_t stride, int64_t band_stride) {
    if (n_bands < 1 || height < 1 || stride < 1 || band_stride < 0) return 0;
    return (uint64_t)job_span(n_bands, height, stride, band_stride) * 5u;
}
// the job's scratch planes (NULL without scratch): offsets (float32) and source flags (1 byte), indexed like the job's planes
static float* job_scratch_offset(const hk_dev_job* job) { return static_cast<float*>(job->scratch); }
static unsigned char* job_scratch_flag(const hk_dev_job* job) {
    return job->scratch ? static_cast<unsigned char*>(job->scratch) +
                              4 * job_span(job->n_bands, job->height, job->stride, job->band_stride)
                        : nullptr;
}

static int check_job(hk_ctx* ctx, const hk_dev_job* job, bool allow_no_rows = false) {
    if (!ctx || !job) return fail(HK_ERR_ARG, "NULL argument");
    if (!job->src || !job->ref) return fail(HK_ERR_ARG, "job src/ref is NULL");
    if (job->n_bands < 1 || job->height < (allow_no_rows ? 0 : 1) || job->width < 1) return fail(HK_ERR_ARG, "empty job");
    if (job->stride < job->width || (job->stride % hk::PX) != 0)
        return fail(HK_ERR_ARG, "job stride must be >= width and a multiple of %d elements", hk::PX);
    if ((job->band_stride % hk::PX) != 0) return fail(HK_ERR_ARG, "band_stride must be a multiple of %d", hk::PX);
    if (((uintptr_t)job->src | (uintptr_t)job->ref | (uintptr_t)job->gain | (uintptr_t)job->offset |
         (uintptr_t)job->r2 | (uintptr_t)job->corr) & 15)
        return fail(HK_ERR_ARG, "device planes must be 16-byte aligned");
    if (job->stream < 0 || job->stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    if (job->scratch) {
        if ((uintptr_t)job->scratch & 15) return fail(HK_ERR_ARG, "job scratch must be 16-byte aligned");
        if (job->scratch_bytes < hk_dev_job_scratch_bytes(job->n_bands, job->height, job->stride, job->band_stride))
            return fail(HK_ERR_ARG, "job scratch smaller than hk_dev_job_scratch_bytes()");
    }
    if (job->out_rows || job->out_cols) {
        if (job->out_row0 < 0 || job->out_col0 < 0 || job->out_rows < 1 || job->out_cols < 1 ||
            job->out_row0 + job->out_rows > job->height || job->out_col0 + job->out_cols > job->width)
            return fail(HK_ERR_ARG, "job store window outside the job");
        if ((job->out_col0 % hk::PX) != 0 || (((job->out_col0 + job->out_cols) % hk::PX) != 0 && job->out_col0 + job->out_cols != job->width))
            return fail(HK_ERR_ARG, "job store window must start and end on multiples of %d columns (or at the job's last column)", hk::PX);
    }
    return HK_OK;
}

// narrow the kernel's store window to the job's (the halo crop of a block processed in place inside a larger raster)
static void apply_job_window(hk::FitArgs& a, const hk_dev_job* job) {
    if (job->out_rows || job->out_cols) {
        a.out_y0 = job->out_row0, a.out_y1 = job->out_row0 + job->out_rows;
        a.out_x0 = job->out_col0, a.out_x1 = job->out_col0 + job->out_cols;
    }
}

int hk_fit_apply_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (desc->model == HK_MODEL_GAIN_BLK_OFFSET && !job->norm) return fail(HK_ERR_ARG, "gain-blk-offset needs job->norm");
    HK_ENTER(ctx);
    hk::FitArgs a;
    memset(&a, 0, sizeof(a));
    a.src = job->src, a.ref = job->ref, a.gain = job->gain, a.offset = job->offset, a.r2 = job->r2, a.corr = job->corr;
    a.norm = job->norm;
    a.fail_count = reinterpret_cast<unsigned long long*>(job->fail_count);
    a.height = job->height, a.width = job->width, a.stride = job->stride, a.band_stride = job->band_stride;
    a.n_bands = job->n_bands;
    fill_args(a, desc, ctx->xcd_remap);
    fill_grid(a, job->seg_rows);
    apply_job_window(a, job);
    if ((job->out_rows || job->out_cols) && a.has_thresh)
        return fail(HK_ERR_UNSUPPORTED, "a store window is not supported together with r2_inpaint_thresh (the in-painting "
                                        "needs the parameters of the whole block)");
    if (desc->model == HK_MODEL_GAIN_OFFSET && a.has_thresh && job->scratch) {
        // a job that carries scratch gets the in-painting's inputs -- offsets + source flags, 5 bytes per pixel -- left there by
        // this pass (hk_inpaint_dev_counts starts from them): a caller that expects pixels to fail the r2 mask provides it
        a.flag = job_scratch_flag(job);
        if (!a.offset) a.offset = job_scratch_offset(job);
    }
    return launch_fit(ctx, ctx->slots[job->stream], a, desc, needs_r2(desc));
}

int hk_fail_counts_async(hk_ctx* ctx, const hk_dev_job* job, uint64_t* host_counts, hk_event* ready) {
    int rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (!job->fail_count || !host_counts || !ready) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    const size_t bytes = (size_t)job->n_bands * sizeof(unsigned long long);
    if (!host_is_pinned(host_counts, bytes)) return fail(HK_ERR_ARG, "host_counts must be page-locked memory (hk_host_alloc / hk_host_register)");
    HK_HIP(hipMemcpyAsync(host_counts, job->fail_count, bytes, hipMemcpyDeviceToHost, sl.stream));
    // the counters are consumed: the next hk_fit_apply_dev into this buffer starts from zero
    HK_HIP(hipMemsetAsync(job->fail_count, 0, bytes, sl.stream));
    HK_HIP(hipEventRecord(ready->ev, sl.stream));
    return HK_OK;
}

int hk_debug_stage_stamps(hk_ctx* ctx, uint64_t out[16], int32_t reset) {
    if (!ctx || !out) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    HK_HIP(hipDeviceSynchronize());
    unsigned long long tmp[16];
    HK_HIP(hk::read_stamps(tmp, reset != 0));
    for (int i = 0; i < 16; ++i) out[i] = tmp[i];
    return HK_OK;
}

int hk_r2_certificate_constants(float thresh, double* pass_below, double* fail_above, float* kappa, float* kappa_fail) {
    if (pass_below) *pass_below = r2_pass_scale(thresh);
    if (fail_above) *fail_above = r2_fail_above(thresh);
    if (kappa) *kappa = r2_fail_scale(thresh);
    if (kappa_fail) *kappa_fail = r2_failcert_scale(thresh);
    return HK_OK;
}

int hk_counts_pending(const uint64_t* counts, int32_t n_bands) {
    if (!counts) return 0;
    for (int32_t b = 0; b < n_bands; ++b)
        if (counts[b]) return 1;  // failing pixels
    return 0;
}

int hk_event_sync(hk_ctx* ctx, hk_event* ev) {
    if (!ctx || !ev) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    HK_HIP(hipEventSynchronize(ev->ev));
    return HK_OK;
}

int hk_inpaint_dev_counts(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, const uint64_t* counts,
                          uint64_t* n_fail_out) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (n_fail_out) *n_fail_out = 0;
    if (desc->model != HK_MODEL_GAIN_OFFSET || !desc->has_r2_thresh) return HK_OK;  // nothing to in-paint
    if (!counts) return fail(HK_ERR_ARG, "counts is NULL");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    const bool r2 = needs_r2(desc);
    const size_t plane = (size_t)job->stride * job->height * sizeof(float);
    unsigned long long total = 0;
    for (int b = 0; b < job->n_bands; ++b) {
        unsigned long long n_fail = counts[b];
        if (n_fail == 0) continue;
        const long long off = (long long)b * job->band_stride;
        hk::FitArgs a;
        memset(&a, 0, sizeof(a));
        a.src = job->src + off, a.ref = job->ref + off;
        a.gain = job->gain ? job->gain + off : nullptr, a.offset = job->offset ? job->offset + off : nullptr;
        a.r2 = job->r2 ? job->r2 + off : nullptr, a.corr = job->corr ? job->corr + off : nullptr;
        a.height = job->height, a.width = job->width, a.stride = job->stride, a.band_stride = 0, a.n_bands = 1;
        fill_args(a, desc, ctx->xcd_remap);
        fill_grid(a, job->seg_rows);
        std::unique_lock<std::mutex> lk(ctx->mu);  // the slot's scratch may be (re)allocated
        // offsets + source flags left by the pass that counted, which writes them whenever the job carries scratch (hk_fit_apply_dev)
        float* pre_off = nullptr;
        const unsigned char* pre_flag = nullptr;
        if (job->scratch) {
            pre_flag = job_scratch_flag(job) + off;
            pre_off = a.offset ? a.offset : job_scratch_offset(job) + off;
        }
        total += n_fail;
        if (n_fail == 0) continue;
        rc = inpaint_band(sl, a, desc, r2, plane, n_fail, false, pre_off, pre_flag);
        if (rc) return rc;
    }
    if (n_fail_out) *n_fail_out = total;
    return HK_OK;
}

int hk_inpaint_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, uint64_t* n_fail_out) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (n_fail_out) *n_fail_out = 0;
    if (desc->model != HK_MODEL_GAIN_OFFSET || !desc->has_r2_thresh) return HK_OK;  // nothing to in-paint
    if (!job->fail_count) return fail(HK_ERR_ARG, "job->fail_count is NULL");
    if (job->n_bands > 1024) return fail(HK_ERR_ARG, "too many bands");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    std::vector<uint64_t> counts((size_t)job->n_bands);
    uint64_t* const pinned = sl.pin<uint64_t>(Slot::PIN_COUNTS);  // <= 1024 counters (checked above)
    HK_HIP(hipMemcpyAsync(pinned, job->fail_count, counts.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream));
    HK_HIP(hipStreamSynchronize(sl.stream));
    memcpy(counts.data(), pinned, counts.size() * sizeof(uint64_t));
    // the counters are consumed: the next hk_fit_apply_dev of this job starts from zero
    HK_HIP(hipMemsetAsync(job->fail_count, 0, counts.size() * sizeof(uint64_t), sl.stream));
    return hk_inpaint_dev_counts(ctx, desc, job, counts.data(), n_fail_out);
}

// Reduction workspace of the device-resident entry points on one stream (block statistics, comparison sums).
static int ensure_stream_ws(hk_ctx* ctx, Slot& sl, size_t need) {
    // device-resident jobs on one stream are issued by one caller at a time (stream order); the lock only protects
    // the (re)allocation against other streams' callers touching the context
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!fits(sl.norm_ws_bytes, need)) {
        if (sl.norm_ws) {
            HK_HIP(hipStreamSynchronize(sl.stream));
            HK_HIP(dev_free(sl.norm_ws));
            sl.norm_ws = nullptr, sl.norm_ws_bytes = 0;
        }
        if (dev_malloc(&sl.norm_ws, need) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", need);
        sl.norm_ws_bytes = need;
    }
    return HK_OK;
}

static int check_nodata_mode(int32_t mode) {
    return (mode == HK_NODATA_NONE || mode == HK_NODATA_NAN || mode == HK_NODATA_VALUE)
               ? HK_OK
               : fail(HK_ERR_ARG, "bad nodata mode %d", mode);
}

int hk_compare_sums_dev(hk_ctx* ctx, const hk_dev_job* job, int32_t src_nodata_mode, float src_nodata,
                        int32_t ref_nodata_mode, float ref_nodata, double* sums_dev) {
    int rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (!sums_dev) return fail(HK_ERR_ARG, "sums_dev is NULL");
    if ((rc = check_nodata_mode(src_nodata_mode)) || (rc = check_nodata_mode(ref_nodata_mode))) return rc;
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    rc = ensure_stream_ws(ctx, sl, hk::compare_workspace_bytes(job->n_bands));
    if (rc) return rc;
    hk::CompareArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.src = job->src, ca.ref = job->ref, ca.height = job->height, ca.width = job->width;
    ca.src_stride = ca.ref_stride = job->stride, ca.src_band_stride = ca.ref_band_stride = job->band_stride;
    ca.n_bands = job->n_bands;
    ca.src_nd_mode = src_nodata_mode, ca.ref_nd_mode = ref_nodata_mode, ca.src_nodata = src_nodata, ca.ref_nodata = ref_nodata;
    HK_HIP(hk::launch_compare_sums(ca, sl.norm_ws, sums_dev, sl.stream));
    return HK_OK;
}

int hk_compare_sums(hk_ctx* ctx, const float* src, int64_t src_stride, int32_t src_nodata_mode, float src_nodata,
                    const float* ref, int64_t ref_stride, int32_t ref_nodata_mode, float ref_nodata, int32_t height,
                    int32_t width, double sums_out[7]) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (!src || !ref || !sums_out) return fail(HK_ERR_ARG, "NULL pointer argument");
    if (height < 1 || width < 1) return fail(HK_ERR_ARG, "empty raster %d x %d", height, width);
    if (src_stride < width || ref_stride < width) return fail(HK_ERR_ARG, "row stride smaller than width");
    int rc;
    if ((rc = check_nodata_mode(src_nodata_mode)) || (rc = check_nodata_mode(ref_nodata_mode))) return rc;
    HK_ENTER(ctx);
    const int64_t stride = (width + ROW_ALIGN - 1) / ROW_ALIGN * ROW_ALIGN;
    const size_t plane = (size_t)stride * height * sizeof(float);
    const size_t ws_bytes = hk::compare_workspace_bytes(1);
    SlotLease lease(ctx);
    Slot& sl = lease.slot();
    rc = ensure_dev(sl, 2 * plane + ws_bytes + 256);
    if (rc) return rc;
    char* base = static_cast<char*>(sl.dev);
    float* d_src = reinterpret_cast<float*>(base);
    float* d_ref = reinterpret_cast<float*>(base + plane);
    void* d_ws = base + 2 * plane;
    double* d_sums = reinterpret_cast<double*>(base + 2 * plane + ws_bytes);
    const size_t wbytes = (size_t)width * sizeof(float);
    if ((rc = stage_h2d(sl, d_src, stride * 4, src, src_stride * 4, wbytes, height))) return rc;
    if ((rc = stage_h2d(sl, d_ref, stride * 4, ref, ref_stride * 4, wbytes, height))) return rc;
    hk::CompareArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.src = d_src, ca.ref = d_ref, ca.height = height, ca.width = width;
    ca.src_stride = ca.ref_stride = stride, ca.src_band_stride = ca.ref_band_stride = 0, ca.n_bands = 1;
    ca.src_nd_mode = src_nodata_mode, ca.ref_nd_mode = ref_nodata_mode, ca.src_nodata = src_nodata, ca.ref_nodata = ref_nodata;
    HK_HIP(hk::launch_compare_sums(ca, d_ws, d_sums, sl.stream));
    HK_HIP(hipMemcpyAsync(sl.pin<double>(Slot::PIN_SUMS), d_sums, 7 * sizeof(double), hipMemcpyDeviceToHost, sl.stream));
    if ((rc = stage_finish(sl))) return rc;
    memcpy(sums_out, sl.pin<double>(Slot::PIN_SUMS), 7 * sizeof(double));
    return HK_OK;
}

int hk_block_norm_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, double* norm_dev) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (!norm_dev) return fail(HK_ERR_ARG, "norm_dev is NULL");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    rc = ensure_stream_ws(ctx, sl, hk::norm_workspace_bytes(job->n_bands, job->height, job->width));
    if (rc) return rc;
    hk::NormArgs na;
    na.src = job->src, na.ref = job->ref, na.height = job->height, na.width = job->width, na.stride = job->stride;
    na.band_stride = job->band_stride, na.n_bands = job->n_bands;
    na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
    na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
    HK_HIP(hk::launch_block_norm(na, sl.norm_ws, norm_dev, sl.stream));
    return HK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched device entry points: many jobs, one launch per kernel stage, all on jobs[0].stream.

// Upload `bytes` of table data to a device buffer of the slot's ring; *dev_out is valid for the kernels queued next on the
// slot's stream (and until the ring comes round: TBL_RING uploads later, which the stream has ordered behind them).
static int upload_table(hk_ctx* ctx, Slot& sl, const void* data, size_t bytes, void** dev_out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int i = sl.tbl_next;
    sl.tbl_next = (sl.tbl_next + 1) % Slot::TBL_RING;
    if (!sl.tbl_ev[i]) HK_HIP(hipEventCreateWithFlags(&sl.tbl_ev[i], hipEventDisableTiming));
    else HK_HIP(hipEventSynchronize(sl.tbl_ev[i]));  // the upload that last used this entry's host buffer has been made
    if (sl.tbl_bytes[i] < bytes) {
        if (sl.tbl_dev[i]) {
            HK_HIP(hipStreamSynchronize(sl.stream));  // kernels may still read the old device buffer
            HK_HIP(dev_free(sl.tbl_dev[i]));
            HK_HIP(hipHostFree(sl.tbl_host[i]));
        }
        sl.tbl_dev[i] = sl.tbl_host[i] = nullptr, sl.tbl_bytes[i] = 0;
        const size_t cap = (bytes + 4095) / 4096 * 4096;
        if (dev_malloc(&sl.tbl_dev[i], cap) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", cap);
        if (hipHostMalloc(&sl.tbl_host[i], cap, hipHostMallocDefault) != hipSuccess) {
            (void)dev_free(sl.tbl_dev[i]);
            sl.tbl_dev[i] = nullptr;
            return fail(HK_ERR_NOMEM, "hipHostMalloc(%zu) failed", cap);
        }
        sl.tbl_bytes[i] = cap;
    }
    memcpy(sl.tbl_host[i], data, bytes);
    HK_HIP(hipMemcpyAsync(sl.tbl_dev[i], sl.tbl_host[i], bytes, hipMemcpyHostToDevice, sl.stream));
    HK_HIP(hipEventRecord(sl.tbl_ev[i], sl.stream));
    *dev_out = sl.tbl_dev[i];
    return HK_OK;
}

static int check_batch(hk_ctx* ctx, const hk_dev_job* jobs, int32_t n_jobs) {
    if (!ctx || !jobs) return fail(HK_ERR_ARG, "NULL argument");
    if (n_jobs < 1 || n_jobs > 65536) return fail(HK_ERR_ARG, "n_jobs %d outside 1..65536", n_jobs);
    for (int32_t j = 0; j < n_jobs; ++j) {
        if (jobs[j].stream != jobs[0].stream) return fail(HK_ERR_ARG, "the jobs of a batch share one stream (job %d differs)", j);
        const int rc = check_job(ctx, &jobs[j]);
        if (rc) return rc;
    }
    return HK_OK;
}

int hk_block_norm_batch_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* jobs, int32_t n_jobs, double* norm_dev) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_batch(ctx, jobs, n_jobs);
    if (rc) return rc;
    DevEnter entered(ctx, jobs[0].stream);
    if (!norm_dev) return fail(HK_ERR_ARG, "norm_dev is NULL");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[jobs[0].stream];
    std::vector<hk::NormPlane> planes;
    int max_h = 0, max_w = 0, max_waves = 0;
    long long max_px = 0;
    for (int32_t j = 0; j < n_jobs; ++j) {
        const hk_dev_job& job = jobs[j];
        for (int32_t b = 0; b < job.n_bands; ++b) {
            hk::NormPlane pl;
            pl.src = job.src + (long long)b * job.band_stride, pl.ref = job.ref + (long long)b * job.band_stride;
            pl.stride = job.stride, pl.height = job.height, pl.width = job.width;
            planes.push_back(pl);
        }
        // the plane with the most pixels sizes the compaction buffers of every plane, the one with the most 1 KB chunks the grid
        // of the streaming pass (not the same plane in general: 1090 x 1025 has fewer pixels but more chunks than 1100 x 1024)
        if ((long long)job.height * job.width > max_px) max_px = (long long)job.height * job.width, max_h = job.height, max_w = job.width;
        max_waves = std::max(max_waves, hk::norm_pass_waves(job.height, job.width));
    }
    // grid.y of the select passes = 2 x planes
    if (planes.size() > 32767) return fail(HK_ERR_ARG, "a statistics batch holds at most 32767 planes (jobs x bands)");
    rc = ensure_stream_ws(ctx, sl, hk::norm_workspace_bytes((int)planes.size(), max_h, max_w));
    if (rc) return rc;
    void* tbl = nullptr;
    rc = upload_table(ctx, sl, planes.data(), planes.size() * sizeof(hk::NormPlane), &tbl);
    if (rc) return rc;
    hk::NormArgs na;
    na.planes = static_cast<const hk::NormPlane*>(tbl);
    na.src = na.ref = nullptr, na.height = max_h, na.width = max_w, na.stride = 0, na.band_stride = 0;
    na.n_bands = (int)planes.size();
    na.grid_waves = max_waves;
    na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
    na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
    HK_HIP(hk::launch_block_norm(na, sl.norm_ws, norm_dev, sl.stream));
    return HK_OK;
}

int hk_fit_apply_batch_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* jobs, int32_t n_jobs) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_batch(ctx, jobs, n_jobs);
    if (rc) return rc;
    DevEnter entered(ctx, jobs[0].stream);
    for (int32_t j = 1; j < n_jobs; ++j) {
        const hk_dev_job* job = &jobs[j];
        // what selects the kernel build must not differ inside a launch (the contract holds for every model)
        if ((!job->gain) != (!jobs[0].gain) || (!job->offset) != (!jobs[0].offset) || (!job->r2) != (!jobs[0].r2) ||
            (!job->corr) != (!jobs[0].corr) || (!job->fail_count) != (!jobs[0].fail_count) || (!job->scratch) != (!jobs[0].scratch))
            return fail(HK_ERR_ARG, "the jobs of a batch ask for the same set of outputs (job %d differs from job 0)", j);
    }
    if (!hk::fit_batch_supported(desc->model, needs_r2(desc))) {
        // builds without the job-table look-up (hk_kernels.h fit_batch_build): one launch per job, in order, same results
        for (int32_t j = 0; j < n_jobs; ++j) {
            rc = hk_fit_apply_dev(ctx, desc, &jobs[j]);
            if (rc) return rc;
        }
        return HK_OK;
    }
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[jobs[0].stream];
    const int wpb = hk::fit_lockstep_waves();
    std::vector<hk::FitJob> table((size_t)n_jobs);
    hk::FitArgs a0;  // the launch's argument block: everything the jobs share, and job 0's own fields (which the table overrides)
    memset(&a0, 0, sizeof(a0));
    long long groups[2] = {0, 0}, total_px = 0;
    bool pad_by_size = false;
    // fill_grid()'s two segment heights, applied to the LAUNCH: the jobs run in order, so the jobs whose units make up the last
    // ~1.25 generations of resident waves get short segments (they level the end of the launch) and all earlier ones long
    // segments (half the priming rows).  Results do not depend on the segment height (exact running sums).
    std::vector<int> seg_of((size_t)n_jobs, 0);
    {
        long long units = 0;
        std::vector<long long> first((size_t)n_jobs + 1, 0);
        int uniform = 0;
        bool any_explicit = false;
        for (int32_t j = 0; j < n_jobs; ++j) {
            hk::FitArgs a;
            memset(&a, 0, sizeof(a));
            a.height = jobs[j].height, a.width = jobs[j].width, a.n_bands = jobs[j].n_bands;
            fill_args(a, desc, ctx->xcd_remap);
            fill_grid(a, jobs[j].seg_rows);
            any_explicit |= jobs[j].seg_rows > 0 || a.seg_rows_pref > 0;
            if (j == 0) uniform = 2 * a.rh + 1 <= 5 ? 64 : (2 * a.rh + 1 <= 9 ? 128 : 256);
            first[(size_t)j] = units;
            units += (long long)a.n_strips * a.n_segs * a.n_bands;
        }
        first[(size_t)n_jobs] = units;
        const long long slots = (long long)env_int("HK_WAVE_SLOTS", 256 * 12);
        if (!any_explicit && units >= 6 * slots) {
            const long long tail_from = units - (long long)(1.25 * (double)slots);
            for (int32_t j = 0; j < n_jobs; ++j) seg_of[(size_t)j] = first[(size_t)j + 1] <= tail_from ? 2 * uniform : uniform / 2;
        }
    }
    for (int32_t j = 0; j < n_jobs; ++j) {
        const hk_dev_job* job = &jobs[j];
        if (desc->model == HK_MODEL_GAIN_BLK_OFFSET && !job->norm) return fail(HK_ERR_ARG, "gain-blk-offset needs job->norm (job %d)", j);
        hk::FitArgs a;
        memset(&a, 0, sizeof(a));
        a.src = job->src, a.ref = job->ref, a.gain = job->gain, a.offset = job->offset, a.r2 = job->r2, a.corr = job->corr;
        a.norm = job->norm;
        a.fail_count = reinterpret_cast<unsigned long long*>(job->fail_count);
        a.height = job->height, a.width = job->width, a.stride = job->stride, a.band_stride = job->band_stride;
        a.n_bands = job->n_bands;
        fill_args(a, desc, ctx->xcd_remap);
        if (j == 0) pad_by_size = a.lds_pad < 0;  // fill_grid() would decide by the job's size: the launch's counts (below)
        fill_grid(a, seg_of[(size_t)j] > 0 ? seg_of[(size_t)j] : job->seg_rows);
        apply_job_window(a, job);
        if ((job->out_rows || job->out_cols) && a.has_thresh)
            return fail(HK_ERR_UNSUPPORTED, "a store window is not supported together with r2_inpaint_thresh (the in-painting "
                                            "needs the parameters of the whole block)");
        // (only models without the r2 mask have builds with the job-table look-up: no certificate, no in-painting inputs here)
        hk::FitJob& e = table[(size_t)j];
        memset(&e, 0, sizeof(e));
        e.src = a.src, e.ref = a.ref, e.gain = a.gain, e.offset = a.offset, e.r2 = a.r2, e.corr = a.corr, e.norm = a.norm;
        e.fail_count = a.fail_count, e.flag = a.flag;
        e.stride = a.stride, e.band_stride = a.band_stride, e.height = a.height, e.width = a.width, e.n_bands = a.n_bands;
        e.seg_rows = a.seg_rows, e.n_strips = a.n_strips, e.n_segs = a.n_segs, e.seg_rows_tail = a.seg_rows_tail;
        e.n_segs_big = a.n_segs_big;
        e.out_y0 = a.out_y0, e.out_y1 = a.out_y1, e.out_x0 = a.out_x0, e.out_x1 = a.out_x1;
        e.first_group[0] = (int)groups[0], e.first_group[1] = (int)groups[1];
        groups[0] += (long long)a.n_strips * a.n_segs * a.n_bands;
        groups[1] += (long long)((a.n_strips + wpb - 1) / wpb) * a.n_segs * a.n_bands;
        total_px += (long long)a.height * a.width * a.n_bands;
        if (groups[0] > 0x7fffff00ll) return fail(HK_ERR_ARG, "the batch has too many wave units for one launch");
        if (j == 0) a0 = a;
    }
    a0.cert_only = 0;
    a0.n_jobs = n_jobs;
    a0.batch_groups[0] = (int)groups[0], a0.batch_groups[1] = (int)groups[1];
    if (pad_by_size) a0.lds_pad = total_px >= (128ll << 20) ? 4096 : 0;  // fill_grid()'s occupancy policy, for the whole launch
    void* tbl = nullptr;
    rc = upload_table(ctx, sl, table.data(), table.size() * sizeof(hk::FitJob), &tbl);
    if (rc) return rc;
    a0.jobs = static_cast<const hk::FitJob*>(tbl);
    HK_HIP(hk::launch_fit_apply(a0, desc->model, needs_r2(desc), sl.stream));
    return HK_OK;
}

int hk_fail_counts_batch_async(hk_ctx* ctx, const hk_dev_job* jobs, int32_t n_jobs, uint64_t* host_counts, hk_event* ready) {
    int rc = check_batch(ctx, jobs, n_jobs);
    if (rc) return rc;
    DevEnter entered(ctx, jobs[0].stream);
    if (!host_counts || !ready) return fail(HK_ERR_ARG, "NULL argument");
    for (int32_t j = 0; j < n_jobs; ++j)
        if (!jobs[j].fail_count) return fail(HK_ERR_ARG, "job %d has no fail_count", j);
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[jobs[0].stream];
    {
        size_t n_all = 0;
        for (int32_t j = 0; j < n_jobs; ++j) n_all += (size_t)jobs[j].n_bands;
        if (!host_is_pinned(host_counts, n_all * sizeof(uint64_t)))
            return fail(HK_ERR_ARG, "host_counts must be page-locked memory (hk_host_alloc / hk_host_register)");
    }
    // one copy + one clearing per run of jobs whose counters lie back to back in device memory (a caller that allocates the
    // counters of a batch as one array gets exactly one of each)
    size_t done = 0;  // counters copied so far = offset into host_counts
    for (int32_t j = 0; j < n_jobs;) {
        uint64_t* const first = jobs[j].fail_count;
        size_t n = (size_t)jobs[j].n_bands;
        int32_t k = j + 1;
        while (k < n_jobs && jobs[k].fail_count == first + n) n += (size_t)jobs[k].n_bands, ++k;
        HK_HIP(hipMemcpyAsync(host_counts + done, first, n * sizeof(uint64_t), hipMemcpyDeviceToHost, sl.stream));
        HK_HIP(hipMemsetAsync(first, 0, n * sizeof(uint64_t), sl.stream));
        done += n, j = k;
    }
    HK_HIP(hipEventRecord(ready->ev, sl.stream));
    return HK_OK;
}

uint64_t hk_block_norm_split_exchange_doubles(int32_t n_bands) {
    return n_bands > 0 ? (uint64_t)hk::norm_split_exchange_doubles(n_bands) : 0;
}

int hk_block_norm_split_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, int32_t phase, int32_t world_size,
                            double* xchg_dev, double* norm_dev) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job, /*allow_no_rows=*/true);  // a rank without rows of the block takes part with zeros
    if (rc) return rc;
    if (!xchg_dev || !norm_dev) return fail(HK_ERR_ARG, "xchg_dev / norm_dev is NULL");
    if (phase < 0 || phase > 5) return fail(HK_ERR_ARG, "phase %d outside 0..5", phase);
    if (world_size < 1) return fail(HK_ERR_ARG, "world_size < 1");
    DevEnter entered(ctx, job->stream);
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    // the phases of one block share the stream's workspace: phase 0 sizes it, the others find it as it was left
    rc = ensure_stream_ws(ctx, sl, hk::norm_workspace_bytes(job->n_bands, job->height > 0 ? job->height : 1, job->width));
    if (rc) return rc;
    hk::NormArgs na;
    na.src = job->src, na.ref = job->ref, na.height = job->height, na.width = job->width, na.stride = job->stride;
    na.band_stride = job->band_stride, na.n_bands = job->n_bands;
    na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
    na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
    HK_HIP(hk::launch_block_norm_split(na, sl.norm_ws, xchg_dev, 1.0 / (double)world_size, phase, norm_dev, sl.stream));
    return HK_OK;
}

int hk_comm_unique_id(uint8_t id[HK_COMM_ID_BYTES]) {
    static_assert(HK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "hk_comm id = ncclUniqueId");
    if (!id) return fail(HK_ERR_ARG, "id is NULL");
    if (!rccl().ok) return fail(HK_ERR_UNSUPPORTED, "librccl could not be opened");
    ncclUniqueId u;
    HK_RCCL(rccl().GetUniqueId(&u));
    memcpy(id, u.internal, HK_COMM_ID_BYTES);
    return HK_OK;
}

int hk_comm_init(hk_ctx* ctx, const uint8_t id[HK_COMM_ID_BYTES], int32_t rank, int32_t world_size) {
    if (!ctx || !id) return fail(HK_ERR_ARG, "NULL argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(HK_ERR_ARG, "rank %d outside a group of %d", rank, world_size);
    if (!rccl().ok) return fail(HK_ERR_UNSUPPORTED, "librccl could not be opened");
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    if (ctx->comm) return fail(HK_ERR_ARG, "the context has a communicator already (hk_comm_destroy it first)");
    HK_ENTER(ctx);
    ncclUniqueId u;
    memcpy(u.internal, id, HK_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    HK_RCCL(rccl().CommInitRank(&c, world_size, u, rank));  // collective: returns once every rank has joined
    ctx->comm = c, ctx->comm_rank = rank, ctx->comm_world = world_size;
    return HK_OK;
}

int hk_comm_destroy(hk_ctx* ctx) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    if (!ctx->comm) return HK_OK;
    HK_ENTER(ctx);
    HK_HIP(hipDeviceSynchronize());
    HK_RCCL(rccl().CommDestroy(ctx->comm));
    ctx->comm = nullptr, ctx->comm_rank = ctx->comm_world = 0;
    return HK_OK;
}

int hk_comm_info(hk_ctx* ctx, int32_t* rank, int32_t* world_size) {
    if (!ctx || !rank || !world_size) return fail(HK_ERR_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    *rank = ctx->comm ? ctx->comm_rank : -1;
    *world_size = ctx->comm ? ctx->comm_world : 0;
    return HK_OK;
}

int hk_comm_allreduce_f64_dev(hk_ctx* ctx, double* buf_dev, uint64_t count, int32_t stream) {
    if (!ctx || !buf_dev) return fail(HK_ERR_ARG, "NULL argument");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    if (!ctx->comm) return fail(HK_ERR_ARG, "the context has no communicator (hk_comm_init)");
    HK_ENTER(ctx);
    HK_RCCL(rccl().AllReduce(buf_dev, buf_dev, count, ncclDouble, ncclSum, ctx->comm, ctx->slots[stream].stream));
    return HK_OK;
}

int hk_block_norm_split_comm_dev(hk_ctx* ctx, const hk_fit_desc* desc, const hk_dev_job* job, double* norm_dev) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    rc = check_job(ctx, job, /*allow_no_rows=*/true);
    if (rc) return rc;
    DevEnter entered(ctx, job->stream);
    if (!norm_dev) return fail(HK_ERR_ARG, "norm_dev is NULL");
    std::lock_guard<std::mutex> lk(ctx->comm_mu);  // one collective sequence at a time per communicator, every rank alike
    if (!ctx->comm) return fail(HK_ERR_ARG, "the context has no communicator (hk_comm_init)");
    HK_ENTER(ctx);
    Slot& sl = ctx->slots[job->stream];
    rc = ensure_stream_ws(ctx, sl, hk::norm_workspace_bytes(job->n_bands, job->height > 0 ? job->height : 1, job->width));
    if (rc) return rc;
    const size_t n = hk::norm_split_exchange_doubles(job->n_bands);
    if (sl.comm_xchg_doubles < n) {
        if (sl.comm_xchg) {
            HK_HIP(hipStreamSynchronize(sl.stream));  // an earlier sequence on this stream may still use it
            HK_HIP(dev_free(sl.comm_xchg));
            sl.comm_xchg = nullptr, sl.comm_xchg_doubles = 0;
        }
        if (dev_malloc(reinterpret_cast<void**>(&sl.comm_xchg), n * sizeof(double)) != hipSuccess)
            return fail(HK_ERR_NOMEM, "hipMalloc(%zu) failed", n * sizeof(double));
        sl.comm_xchg_doubles = n;
    }
    hk::NormArgs na;
    na.src = job->src, na.ref = job->ref, na.height = job->height, na.width = job->width, na.stride = job->stride;
    na.band_stride = job->band_stride, na.n_bands = job->n_bands;
    na.src_nd_mode = desc->src_nodata_mode, na.ref_nd_mode = desc->ref_nodata_mode;
    na.src_nodata = desc->src_nodata, na.ref_nodata = desc->ref_nodata;
    // six phases on the slab, five all-reduces between them, all queued on the job's stream: no host synchronisation
    for (int phase = 0; phase < 6; ++phase) {
        HK_HIP(hk::launch_block_norm_split(na, sl.norm_ws, sl.comm_xchg, 1.0 / (double)ctx->comm_world, phase, norm_dev, sl.stream));
        if (phase < 5)
            HK_RCCL(rccl().AllReduce(sl.comm_xchg, sl.comm_xchg, n, ncclDouble, ncclSum, ctx->comm, sl.stream));
    }
    return HK_OK;
}

int hk_synth_fill_dev(hk_ctx* ctx, float* src, float* ref, int32_t n_bands, int32_t height, int32_t width,
                      int64_t stride, int64_t band_stride, uint64_t seed, int32_t nodata_variant, int32_t stream) {
    if (!ctx || !src || !ref) return fail(HK_ERR_ARG, "NULL argument");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_ENTER(ctx);
    HK_HIP(hk::launch_synth_fill(src, ref, n_bands, height, width, stride, band_stride, seed, nodata_variant,
                                 ctx->slots[stream].stream));
    return HK_OK;
}

int hk_stream_probe_dev(hk_ctx* ctx, const void* a, const void* b, void* out, size_t n_bytes, int32_t stream) {
    if (!ctx || !a || !b || !out) return fail(HK_ERR_ARG, "NULL argument");
    if (n_bytes % 16) return fail(HK_ERR_ARG, "n_bytes must be a multiple of 16");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_ENTER(ctx);
    HK_HIP(hk::launch_stream_probe(a, b, out, n_bytes, ctx->slots[stream].stream));
    return HK_OK;
}

int hk_debug_checksum_dev(hk_ctx* ctx, const float* plane, int64_t stride, int32_t height, int32_t width, int32_t stream,
                          uint64_t* sum_out) {
    if (!ctx || !plane || !sum_out) return fail(HK_ERR_ARG, "NULL argument");
    if (height < 1 || width < 1 || stride < width) return fail(HK_ERR_ARG, "bad window %d x %d, stride %lld", height, width, (long long)stride);
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_ENTER(ctx);
    DevEnter enter(ctx, stream);
    Slot& sl = ctx->slots[stream];
    // the slot's pinned counter word doubles as the device-visible accumulator's landing place
    unsigned long long* d_acc = nullptr;
    if (dev_malloc(reinterpret_cast<void**>(&d_acc), sizeof(unsigned long long)) != hipSuccess) return fail(HK_ERR_NOMEM, "hipMalloc(8) failed");
    hipError_t e = hipMemsetAsync(d_acc, 0, sizeof(unsigned long long), sl.stream);
    if (e == hipSuccess) e = hk::launch_checksum(plane, stride, height, width, d_acc, sl.stream);
    unsigned long long host = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(sl.fail_host, d_acc, sizeof(host), hipMemcpyDeviceToHost, sl.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(sl.stream);
    if (e == hipSuccess) host = *sl.fail_host;
    (void)dev_free(d_acc);
    if (e != hipSuccess) return fail(HK_ERR_HIP, "checksum: %s", hipGetErrorString(e));
    *sum_out = host;
    return HK_OK;
}

int hk_event_create(hk_ctx* ctx, hk_event** ev) {
    if (!ctx || !ev) return fail(HK_ERR_ARG, "NULL argument");
    HK_ENTER(ctx);
    hk_event* e = new (std::nothrow) hk_event();
    if (!e) return fail(HK_ERR_NOMEM, "out of host memory");
    hipError_t he = hipEventCreate(&e->ev);
    if (he != hipSuccess) {
        delete e;
        return fail(HK_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(he));
    }
    *ev = e;
    return HK_OK;
}
int hk_event_destroy(hk_ctx* ctx, hk_event* ev) {
    if (!ctx || !ev) return HK_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return HK_OK;
}
int hk_event_record(hk_ctx* ctx, hk_event* ev, int32_t stream) {
    if (!ctx || !ev) return fail(HK_ERR_ARG, "NULL argument");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_HIP(hipEventRecord(ev->ev, ctx->slots[stream].stream));
    return HK_OK;
}
int hk_stream_wait_event(hk_ctx* ctx, int32_t stream, hk_event* ev) {
    if (!ctx || !ev) return fail(HK_ERR_ARG, "NULL argument");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_ENTER(ctx);
    HK_HIP(hipStreamWaitEvent(ctx->slots[stream].stream, ev->ev, 0));
    return HK_OK;
}
int hk_event_elapsed_ms(hk_ctx* ctx, hk_event* start, hk_event* stop, float* ms) {
    if (!ctx || !start || !stop || !ms) return fail(HK_ERR_ARG, "NULL argument");
    HK_HIP(hipEventSynchronize(stop->ev));
    HK_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return HK_OK;
}
int hk_stream_sync(hk_ctx* ctx, int32_t stream) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    if (stream < 0 || stream >= (int)ctx->slots.size()) return fail(HK_ERR_ARG, "bad stream index");
    HK_HIP(hipStreamSynchronize(ctx->slots[stream].stream));
    return HK_OK;
}

int hk_selftest(hk_ctx* ctx) {
    if (!ctx) return fail(HK_ERR_ARG, "ctx is NULL");
    HK_ENTER(ctx);
    int* d = nullptr;
    HK_HIP(dev_malloc(reinterpret_cast<void**>(&d), sizeof(int)));
    int code = -1;
    // on the transfer slot, one caller at a time: the pooled streams and their pinned words belong to leased / device-job calls
    // (PIN_WORD of slot 0 is where hk_inpaint_dev_counts reads a band's failure count from)
    std::lock_guard<std::mutex> lk(ctx->xfer_mu);
    Slot& xs = ctx->xfer;
    hipError_t e = hipMemsetAsync(d, 0, sizeof(int), xs.stream);
    if (e == hipSuccess) e = hk::launch_selftest(d, xs.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(xs.pin<int>(Slot::PIN_WORD), d, sizeof(int), hipMemcpyDeviceToHost, xs.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(xs.stream);
    if (e == hipSuccess) code = *xs.pin<int>(Slot::PIN_WORD);
    (void)dev_free(d);  // on every path
    if (e != hipSuccess) return fail(HK_ERR_HIP, "self-test launch failed: %s", hipGetErrorString(e));
    if (code != 0) return fail(HK_ERR_HIP, "cross-lane self-test failed (code 0x%x)", code);
    return HK_OK;
}

}  // extern "C"

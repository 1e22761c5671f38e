// csrc/common.cpp — error reporting, device / pinned workspaces, context lifetime.
#include <dlfcn.h>

#include "common.h"
#ifdef MRGFE_TESTING
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#endif
#include "ingest.h"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>

namespace mrgfe {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

bool poll_disabled()
{
    static const bool off = [] { const char* e = std::getenv("MRGFE_NO_POLL"); return e && e[0] == '1'; }();
    return off;
}

// MRGFE_POISON=1 (tests): every fresh device / pinned allocation is filled with 0xCD — fresh memory from the driver is usually zero, which hides a kernel
// that reads what nobody wrote
static bool poison_allocations() { static const bool v = std::getenv("MRGFE_POISON") != nullptr; return v; }

// ---- allocation-failure injector (hardening tests; -DMRGFE_TESTING builds only: libmrgfe_testing.so) ---------------------------------
// mrgfe_dbg_fail_alloc_after(k) / MRGFE_FAIL_ALLOC_AFTER=k: the k-th device / pinned allocation from now (0 = the next one) reports an
// out-of-memory error instead of calling HIP; every later one works again.  tests/faultinject/ sweeps k over whole calls —
// mrgfe_batch_align, mrgfe_prefilter, mrgfe_map_store_generate — and wants an error code from each, no leak, no std::terminate, and a
// correct answer from the next call.  The shipped library compiles the check away.
#ifdef MRGFE_TESTING
static std::atomic<long> g_fail_alloc_in{[] { const char* e = std::getenv("MRGFE_FAIL_ALLOC_AFTER"); return e ? std::atol(e) : -1L; }()};
static std::atomic<long> g_allocs{0};
static bool inject_alloc_failure()
{
    g_allocs.fetch_add(1, std::memory_order_relaxed);
    long v = g_fail_alloc_in.load(std::memory_order_relaxed);
    while (v >= 0) {
        if (g_fail_alloc_in.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {
            if (v == 0) { set_error("out of memory (injected: MRGFE_FAIL_ALLOC_AFTER / mrgfe_dbg_fail_alloc_after)"); return true; }
            return false;
        }
    }
    return false;
}
long fail_alloc_after(long k)
{
    g_fail_alloc_in.store(k, std::memory_order_relaxed);
    return g_allocs.exchange(0, std::memory_order_relaxed);
}
// A process that ends in abort() under the injector (std::terminate of a helper thread, the runtime's own abort) says where: the native stack of the aborting
// thread on stderr, then whatever handler was there before (pytest's faulthandler prints the Python side).  TESTING library only.
static struct sigaction g_prev_abrt;
static void abrt_backtrace(int sig, siginfo_t* info, void* uc)
{
    static const char head[] = "\n[mrgfe testing] SIGABRT; native stack of the aborting thread:\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    if (g_prev_abrt.sa_flags & SA_SIGINFO) { if (g_prev_abrt.sa_sigaction) g_prev_abrt.sa_sigaction(sig, info, uc); }
    else if (g_prev_abrt.sa_handler != SIG_DFL && g_prev_abrt.sa_handler != SIG_IGN && g_prev_abrt.sa_handler) g_prev_abrt.sa_handler(sig);
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}
static const bool g_abrt_installed = [] {
    struct sigaction sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = abrt_backtrace;
    sa.sa_flags = SA_SIGINFO;
    sigemptyset(&sa.sa_mask);
    return sigaction(SIGABRT, &sa, &g_prev_abrt) == 0;
}();
#else
static inline bool inject_alloc_failure() { return false; }
#endif

// ---- roctx ranges -------------------------------------------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        if (std::getenv("MRGFE_NO_ROCTX")) return;
        for (const char* name : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so"}) {
            if (void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
TraceRange::TraceRange(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
TraceRange::~TraceRange() { if (on) roctx().pop(); }

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return MRGFE_OK;
    size_t want = cap ? cap : 4096;
    while (want < bytes) want += want / 2 + 4096;
    want = (want + 255) & ~size_t(255);
    if (p) { MRGFE_HIP_CHECK(hipFree(p)); p = nullptr; cap = 0; }
    if (inject_alloc_failure()) return MRGFE_ERR_HIP;
    MRGFE_HIP_CHECK(hipMalloc(&p, want));
    cap = want;
    if (poison_allocations()) { MRGFE_HIP_CHECK(hipMemset(p, 0xCD, want)); MRGFE_HIP_CHECK(hipDeviceSynchronize()); }
    return MRGFE_OK;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int PinBuf::ensure(size_t bytes)
{
    if (bytes <= cap) return MRGFE_OK;
    size_t want = cap ? cap : 4096;
    while (want < bytes) want += want / 2 + 4096;
    if (p) { MRGFE_HIP_CHECK(hipHostFree(p)); p = nullptr; cap = 0; }
    if (inject_alloc_failure()) return MRGFE_ERR_HIP;
    MRGFE_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    if (poison_allocations()) std::memset(p, 0xCD, want);
    return MRGFE_OK;
}
void PinBuf::release()
{
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

int Arena::alloc(size_t bytes, void** out)
{
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes == 0) bytes = 256;
    for (auto& c : chunks)
        if (c.cap - c.used >= bytes) {
            *out = static_cast<char*>(c.p) + c.used;
            c.used += bytes;
            return MRGFE_OK;
        }
    Chunk c;
    c.cap = bytes > chunk_bytes ? bytes : chunk_bytes;
    c.used = 0;
    c.p = nullptr;
    if (inject_alloc_failure()) return MRGFE_ERR_HIP;
    MRGFE_HIP_CHECK(hipMalloc(&c.p, c.cap));
    if (poison_allocations()) { MRGFE_HIP_CHECK(hipMemset(c.p, 0xCD, c.cap)); MRGFE_HIP_CHECK(hipDeviceSynchronize()); }
    *out = c.p;
    c.used = bytes;
    chunks.push_back(c);
    return MRGFE_OK;
}
void Arena::reset()
{
    for (auto& c : chunks) c.used = 0;
}
void Arena::release()
{
    for (auto& c : chunks) (void)hipFree(c.p);
    chunks.clear();
}

namespace {
// A handful of persistent workers.  Between the rounds of an alignment batch they SPIN on the epoch counter (a round
// comes every millisecond or so and a condition-variable wake-up costs 30-60 us on a large host, as much as the work
// it hands out); outside host_parallel_hot(true) .. (false) they sleep on the condition variable.
class HostPool {
   public:
    HostPool()
    {
        int n = 8;  // total threads, the caller included
        if (const char* e = std::getenv("MRGFE_HOST_THREADS")) n = std::atoi(e);
        unsigned hw = std::thread::hardware_concurrency();
        // one process per GPU (torchrun sets LOCAL_WORLD_SIZE): the ranks of a node share its cores, and these threads spin
        if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
            const int ranks = std::atoi(e);
            if (hw && ranks > 1) hw = std::max(1u, hw / static_cast<unsigned>(ranks));
        }
        if (hw && static_cast<unsigned>(n) > hw) n = static_cast<int>(hw);
        n = n < 1 ? 1 : (n > 8 ? 8 : n);
        for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this, i] { loop(i); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true);
            epoch_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void set_hot(bool hot)  // counted: several contexts may be inside a batch at once
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            hot_.fetch_add(hot ? 1 : -1, std::memory_order_release);
        }
        if (hot) cv_.notify_all();
    }
    void run(int n, const std::function<void(int, int)>& body)
    {
        const int parts = static_cast<int>(workers_.size()) + 1;
        const int chunk = (n + parts - 1) / parts;
        body_ = &body;
        n_ = n;
        chunk_ = chunk;
        pending_.store(static_cast<int>(workers_.size()), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);  // pairs with the predicate check of a sleeping worker
            epoch_.fetch_add(1, std::memory_order_release);
        }
        if (hot_.load(std::memory_order_relaxed) <= 0) cv_.notify_all();
        body(0, chunk < n ? chunk : n);  // the caller takes the first chunk
        while (pending_.load(std::memory_order_acquire) != 0) cpu_relax();
        body_ = nullptr;
    }
    int size() const { return static_cast<int>(workers_.size()) + 1; }

   private:
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            while (epoch_.load(std::memory_order_acquire) == seen) {
                if (hot_.load(std::memory_order_relaxed) > 0) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return epoch_.load(std::memory_order_acquire) != seen || hot_.load(std::memory_order_relaxed) > 0; });
            }
            seen = epoch_.load(std::memory_order_acquire);
            if (stop_.load()) return;
            const std::function<void(int, int)>* body = body_;
            const int b = (id + 1) * chunk_, e = std::min(n_, (id + 2) * chunk_);
            if (body && b < e) (*body)(b, e);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(int, int)>* body_ = nullptr;
    int n_ = 0, chunk_ = 0;
    std::atomic<int>      pending_{0};
    std::atomic<uint64_t> epoch_{0};
    std::atomic<bool>     stop_{false};
    std::atomic<int>      hot_{0};
};
std::mutex g_pool_mu;  // one parallel region at a time (contexts on different GPUs share the pool)
HostPool& host_pool()
{
    static HostPool pool;
    return pool;
}
}  // namespace

void host_parallel_for(int n, int min_serial, const std::function<void(int, int)>& body)
{
    if (n <= 0) return;
    if (n < min_serial) { body(0, n); return; }
    HostPool& pool = host_pool();
    if (pool.size() == 1) { body(0, n); return; }
    std::unique_lock<std::mutex> lk(g_pool_mu, std::try_to_lock);
    if (!lk.owns_lock()) { body(0, n); return; }  // another context is inside a parallel region: do not wait for it
    pool.run(n, body);
}

void host_parallel_hot(bool hot) { host_pool().set_hot(hot); }

int decode_layout(size_t layout, uint32_t* stride, uint32_t* xyz_off, int32_t* intensity_off)
{
    if (layout == 0) layout = 16;
    const uint32_t st = static_cast<uint32_t>(layout & 0xFFFFu), io1 = static_cast<uint32_t>((layout >> 16) & 0xFFu), xo = static_cast<uint32_t>((layout >> 24) & 0xFFu);
    if ((layout >> 32) != 0 || st < 16 || (st % 4) != 0) { set_error("point stride must be a multiple of 4 and >= 16 bytes (got layout 0x%zx)", layout); return MRGFE_ERR_INVALID; }
    if (io1 == 0) {
        // a bare stride says nothing about where the intensity lives.  16 is the packed x,y,z,intensity record; anything wider
        // must name its layout (a bare 32 used to read pcl::PointXYZI's 1.0f padding word as the intensity)
        if (st != 16 || xo != 0) {
            set_error("stride_bytes %u without a layout: pass MRGFE_LAYOUT(stride, xyz_offset, intensity_offset), e.g. MRGFE_LAYOUT_PCL_XYZI for pcl::PointXYZI", st);
            return MRGFE_ERR_INVALID;
        }
        *stride = 16; *xyz_off = 0; *intensity_off = 12;
        return MRGFE_OK;
    }
    const uint32_t io = io1 - 1;
    if ((io % 4) != 0 || (xo % 4) != 0 || io + 4 > st || xo + 12 > st || (io + 4 > xo && io < xo + 12)) {
        set_error("layout 0x%zx: xyz offset %u / intensity offset %u do not fit a %u-byte point", layout, xo, io, st);
        return MRGFE_ERR_INVALID;
    }
    *stride = st; *xyz_off = xo; *intensity_off = static_cast<int32_t>(io);
    return MRGFE_OK;
}

int upload_cloud(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t layout, void* d_dst, int pin_slot)
{
    if (n == 0) return MRGFE_OK;
    uint32_t stride = 16, xo = 0;
    int32_t  io = 12;
    MRGFE_TRY(decode_layout(layout, &stride, &xo, &io));
    (void)pin_slot;
    if (!(stride == 16 && xo == 0 && io == 12))  // strided records: raw bytes up in one copy, x/y/z/intensity gathered on the device (ingest.hip)
        return upload_gathered(ctx, xyzi, n * size_t(stride), n, static_cast<uint32_t>(n), static_cast<uint32_t>(n) * stride, stride, xo, xo + 4, xo + 8, io, d_dst);
    // mrgfe_ctx_set_zero_copy_uploads(ctx, 1): a cloud that sits in page-locked host memory (mrgfe_pin_host_buffer, hipHostMalloc, hipHostRegister)
    // goes up by DMA straight out of the caller's buffer: the staging copy below is a single-thread memcpy (~25 GB/s) and halves what the link gives.
    // The copy is stream-ordered like every upload, so under that switch such a buffer must stay unchanged until the call that consumes the cloud
    // has returned (mrgfe.h) — the reference's keyframe clouds are immutable ConstPtr clouds.  Off by default: the add / set call is done with the
    // caller's memory when it returns.
    if (ctx->zero_copy_uploads && n * 16 >= (size_t(64) << 10)) {
        // BOTH ends of the cloud must lie in page-locked memory (a cloud whose tail leaves the registered range would be DMA'd out of a
        // partly pageable buffer); a registration covers whole pages, so two page-locked ends with a pageable hole between them would take two
        // registrations inside one cloud — not something a caller that pins its clouds produces
        hipPointerAttribute_t a0, a1;
        const char* last = reinterpret_cast<const char*>(xyzi) + n * 16 - 1;
        if (hipPointerGetAttributes(&a0, xyzi) == hipSuccess && a0.type == hipMemoryTypeHost &&
            hipPointerGetAttributes(&a1, last) == hipSuccess && a1.type == hipMemoryTypeHost) {
            MRGFE_HIP_CHECK(hipMemcpyAsync(d_dst, xyzi, n * 16, hipMemcpyHostToDevice, ctx->stream));
            ctx->dma_from_caller = true;  // a failing consuming call waits for the stream before it hands the caller's buffers back (drain_caller_dma)
            return MRGFE_OK;
        }
        (void)hipGetLastError();  // (an unregistered pointer is an error to the query, not to us)
    }
    const int slot = ctx->up_next;
    ctx->up_next ^= 1;
    if (!ctx->up_ev[slot]) MRGFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->up_ev[slot], hipEventDisableTiming));
    // this staging buffer may still be the source of the copy issued two uploads ago
    if (ctx->up_busy[slot]) { MRGFE_HIP_CHECK(hipEventSynchronize(ctx->up_ev[slot])); ctx->up_busy[slot] = false; }
    PinBuf& pb = ctx->up_pin[slot];
    MRGFE_TRY(pb.ensure(n * 16));
    float* dst = pb.as<float>();
    std::memcpy(dst, xyzi, n * 16);
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_dst, dst, n * 16, hipMemcpyHostToDevice, ctx->stream));
    MRGFE_HIP_CHECK(hipEventRecord(ctx->up_ev[slot], ctx->stream));
    ctx->up_busy[slot] = true;
    return MRGFE_OK;  // stream-ordered: later work on ctx->stream sees the cloud; the caller's buffer is already free
}

void drain_caller_dma(mrgfe_ctx* ctx)
{
    // error paths only: a zero-copy upload may still be reading the caller's page-locked buffer when a consuming call gives up — the header promises
    // the buffer is the caller's again once that call has returned, so the stream is waited for first
    if (!ctx || !ctx->dma_from_caller) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->dma_from_caller = false;
}

}  // namespace mrgfe

int mrgfe_ctx::bind()
{
    MRGFE_HIP_CHECK(hipSetDevice(device));
    return MRGFE_OK;
}

int mrgfe_ctx::stage_h2d(void* d_dst, const void* src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return MRGFE_OK;
    const int slot = stage_next;
    stage_next = (stage_next + 1) % kStageSlots;
    if (!stage_ev[slot]) MRGFE_HIP_CHECK(hipEventCreateWithFlags(&stage_ev[slot], hipEventDisableTiming));
    if (stage_busy[slot]) { MRGFE_HIP_CHECK(hipEventSynchronize(stage_ev[slot])); stage_busy[slot] = false; }
    MRGFE_TRY(stage_pin[slot].ensure(bytes));
    std::memcpy(stage_pin[slot].p, src, bytes);
    MRGFE_HIP_CHECK(hipMemcpyAsync(d_dst, stage_pin[slot].p, bytes, hipMemcpyHostToDevice, st));
    MRGFE_HIP_CHECK(hipEventRecord(stage_ev[slot], st));
    stage_busy[slot] = true;
    return MRGFE_OK;
}

extern "C" {

const char* mrgfe_last_error(void) { return mrgfe::get_error(); }
const char* mrgfe_version(void) { return "mrgfe 0.1 (gfx950)"; }

static hipError_t create_stream(hipStream_t* st, int priority)  // > 0: the device's highest stream priority, < 0: its lowest, 0: the default
{
    if (priority == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    int least = 0, greatest = 0;  // numerically lower = higher priority
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, priority > 0 ? greatest : least);
}

}  // extern "C"

int mrgfe_ctx::make_stream(hipStream_t* st) const
{
    const hipError_t e = cu_mask.empty() ? create_stream(st, priority)
                                         : hipExtStreamCreateWithCUMask(st, static_cast<uint32_t>(cu_mask.size()), cu_mask.data());
    if (e != hipSuccess) { mrgfe::set_error("stream creation failed: %s", hipGetErrorString(e)); return MRGFE_ERR_HIP; }
    return MRGFE_OK;
}

extern "C" {

static int ctx_create(int device_id, int high_priority, int reserve_cus, const mrgfe_ctx* like, mrgfe_ctx** out);

int mrgfe_ctx_create_priority(int device_id, int high_priority, mrgfe_ctx** out) { return ctx_create(device_id, high_priority, 0, nullptr, out); }

int mrgfe_ctx_create_reserving(int device_id, int reserve_cus, mrgfe_ctx** out)
{
    if (reserve_cus < 0 && reserve_cus != MRGFE_RESERVE_AUTO) { mrgfe::set_error("mrgfe_ctx_create_reserving: reserve_cus must not be negative"); return MRGFE_ERR_INVALID; }
    return ctx_create(device_id, 0, reserve_cus, nullptr, out);
}


static int ctx_create(int device_id, int high_priority, int reserve_cus, const mrgfe_ctx* like, mrgfe_ctx** out)
{
    if (!out) { mrgfe::set_error("mrgfe_ctx_create: out is NULL"); return MRGFE_ERR_INVALID; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        mrgfe::set_error("no HIP device available (%s): libmrgfe has no CPU fallback", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return MRGFE_ERR_HIP;
    }
    if (device_id < 0 || device_id >= count) { mrgfe::set_error("device %d out of range (0..%d)", device_id, count - 1); return MRGFE_ERR_INVALID; }
    mrgfe_ctx* c = new mrgfe_ctx();
    c->device = device_id;
    if (c->bind() != MRGFE_OK) { delete c; return MRGFE_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if (like) {
        c->cu_mask = like->cu_mask;
        c->cu_count = like->cu_count;
        c->zero_copy_uploads = like->zero_copy_uploads;
    } else if (reserve_cus == MRGFE_RESERVE_AUTO && c->cu_count < 8) {
        // nothing sensible to split off: a plain context
    } else if (reserve_cus > 0 || reserve_cus == MRGFE_RESERVE_AUTO) {
        if (reserve_cus == MRGFE_RESERVE_AUTO) reserve_cus = std::min(64, c->cu_count / 4);
        // the LAST reserve_cus bits of the mask stay clear: the kernels of this context never occupy those compute units, so the small launches of
        // other contexts (a robot's per-scan path beside a loop-closure batch) always find them free
        const int total = c->cu_count;
        if (reserve_cus >= total) { mrgfe::set_error("mrgfe_ctx_create_reserving: %d of %d compute units reserved leaves none", reserve_cus, total); delete c; return MRGFE_ERR_INVALID; }
        c->cu_mask.assign((total + 31) / 32, 0u);
        for (int k = 0; k < total - reserve_cus; ++k) c->cu_mask[k / 32] |= 1u << (k % 32);
        // (cu_count stays the device's: it enters the tiles-per-item of the derivative launches, i.e. the grouping of their partial sums — a context's
        // mask must not change a bit of any result)
    }
    c->priority = high_priority;
    const bool stream_ok = c->make_stream(&c->stream) == MRGFE_OK;
    if (!stream_ok || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        mrgfe::set_error("failed to create HIP stream / events");
        delete c;
        return MRGFE_ERR_HIP;
    }
    for (int m = 0; m < 3; ++m)
        for (int k = 0; k < 2; ++k)
            if (hipEventCreate(&c->ev_mode[m][k]) != hipSuccess) { mrgfe::set_error("failed to create HIP events"); delete c; return MRGFE_ERR_HIP; }
    *out = c;
    return MRGFE_OK;
}

int mrgfe_ctx_create(int device_id, mrgfe_ctx** out) { return mrgfe_ctx_create_priority(device_id, 0, out); }

void mrgfe_ctx_destroy(mrgfe_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    mrgfe::ctx_tmp_grid_free(ctx);
    for (auto& b : ctx->pf_buf) b.release();
    ctx->pf_state.release();
    ctx->pf_status.release();
    for (auto& b : ctx->scratch) b.release();
    ctx->sort_chunks.release();
    for (auto& b : ctx->pin) b.release();
    for (auto& b : ctx->up_pin) b.release();
    ctx->up_raw.release();
    ctx->up_out.release();
    for (auto& e : ctx->up_ev) if (e) (void)hipEventDestroy(e);
    for (auto& b : ctx->stage_pin) b.release();
    for (auto& e : ctx->stage_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (auto& pr : ctx->ev_mode) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_fit) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->ev_side) if (e) (void)hipEventDestroy(e);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    for (auto& e : ctx->knn_stats.ev) if (e) (void)hipEventDestroy(e);
    ctx->knn_stats.counter.release();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mrgfe_pin_host_buffer(mrgfe_ctx* ctx, void* p, size_t bytes)
{
    if (!ctx || !p || bytes == 0) { mrgfe::set_error("mrgfe_pin_host_buffer: NULL context / pointer or empty range"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx->bind());
    MRGFE_HIP_CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return MRGFE_OK;
}

int mrgfe_ctx_set_zero_copy_uploads(mrgfe_ctx* ctx, int on)
{
    if (!ctx) { mrgfe::set_error("mrgfe_ctx_set_zero_copy_uploads: NULL context"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);  // (an asynchronous align of a batch on this context holds the lock: the switch takes effect behind it)
    ctx->zero_copy_uploads = on != 0;
    return MRGFE_OK;
}

int mrgfe_unpin_host_buffer(mrgfe_ctx* ctx, void* p)
{
    if (!ctx || !p) { mrgfe::set_error("mrgfe_unpin_host_buffer: NULL context / pointer"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx->bind());
    MRGFE_HIP_CHECK(hipHostUnregister(p));
    return MRGFE_OK;
}

int mrgfe_ctx_synchronize(mrgfe_ctx* ctx)
{
    if (!ctx) { mrgfe::set_error("NULL context"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(ctx->bind());
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MRGFE_OK;
}

void* mrgfe_ctx_stream(mrgfe_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

int mrgfe_ctx_knn_stats(mrgfe_ctx* ctx, double out[5])
{
    if (!ctx || !out) { mrgfe::set_error("mrgfe_ctx_knn_stats: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    mrgfe::KnnStats& ks = ctx->knn_stats;
    for (int k = 0; k < 5; ++k) out[k] = 0.0;
    if (!ks.launches || !ks.ev[1]) return MRGFE_OK;
    MRGFE_HIP_CHECK(hipEventSynchronize(ks.ev[1]));
    float ms = 0;
    MRGFE_HIP_CHECK(hipEventElapsedTime(&ms, ks.ev[0], ks.ev[1]));
    unsigned long long cand = 0;
    if (ks.counted) MRGFE_HIP_CHECK(hipMemcpy(&cand, ks.counter.p, 8, hipMemcpyDeviceToHost));
    out[0] = ms;
    out[1] = double(ks.queries);
    out[2] = double(ks.k);
    out[3] = double(cand);
    out[4] = double(ks.launches);
    return MRGFE_OK;
}

int mrgfe_ctx_fitness_stats(mrgfe_ctx* ctx, double out[11])
{
    if (!ctx || !out) { mrgfe::set_error("mrgfe_ctx_fitness_stats: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    const mrgfe::FitStats& f = ctx->fit_stats;
    const double v[11] = {f.ms_block, f.ms_sweep, f.ms_far, double(f.queries), double(f.queued), double(f.queued_far), double(f.words), double(f.tested), double(f.cells), double(f.points), double(f.calls)};
    std::memcpy(out, v, sizeof(v));
    return MRGFE_OK;
}

}  // extern "C"

namespace mrgfe {
int ctx_create_like(const mrgfe_ctx* parent, mrgfe_ctx** out, int priority)
{
    if (!parent) { set_error("ctx_create_like: parent is NULL"); return MRGFE_ERR_INVALID; }
    return ctx_create(parent->device, priority, 0, parent, out);
}
}  // namespace mrgfe

// csrc/common.h — host-side plumbing shared by the translation units of libmrgfe.so:
// error reporting, the per-GPU context (stream + grow-only device/pinned workspaces) and a bump arena.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrgfe.h"
#include "../../include/mrgfe_debug.h"

namespace mrgfe {

void set_error(const char* fmt, ...);
#ifdef MRGFE_TESTING
long fail_alloc_after(long k);  // allocation-failure injector (common.cpp): the k-th allocation from now fails (k < 0: off); returns the allocations counted since the last call
#endif

#define MRGFE_HIP_CHECK(expr)                                                                                   \
    do {                                                                                                        \
        hipError_t _e = (expr);                                                                                 \
        if (_e != hipSuccess) {                                                                                 \
            ::mrgfe::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return MRGFE_ERR_HIP;                                                                               \
        }                                                                                                       \
    } while (0)

#define MRGFE_TRY(expr)              \
    do {                             \
        int _s = (expr);             \
        if (_s != MRGFE_OK) return _s; \
    } while (0)

// grow-only device buffer (avoids a hipMalloc per scan: SURVEY.md §8b "Ownership")
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    int    ensure(size_t bytes);
    void   release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// grow-only pinned host buffer
struct PinBuf {
    void*  p = nullptr;
    size_t cap = 0;
    int    ensure(size_t bytes);
    void   release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// Wait for a record a kernel writes into pinned host memory (system-scope release store of its last word): `done()` reads it.  A bounded spin with
// a pause instruction between the loads (the poll saves ~10 us per round over a stream wait, which is what a single registration's latency is made
// of) that asks the stream now and then, so that a failed launch cannot hang the caller; past the budget (~0.3 ms: a core shared with the other nodes
// of a robot is not held for a long kernel) it falls back to hipStreamSynchronize.  MRGFE_NO_POLL=1: no spinning at all, every wait is a stream wait.
bool poll_disabled();
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}
template <class Done>
inline int poll_host_record(hipStream_t st, Done done, const char* what)
{
    if (!poll_disabled()) {
        for (uint32_t spin = 0; spin < (1u << 16); ++spin) {  // ~65k paused loads: a few hundred microseconds
            if (done()) return MRGFE_OK;
            cpu_relax();
            if ((spin & 0x3ff) != 0x3ff) continue;
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {  // everything queued has run: the record is there, or never will be
                if (done()) return MRGFE_OK;
                set_error("%s: the kernel did not report", what);
                return MRGFE_ERR_HIP;
            }
            if (q != hipErrorNotReady) { set_error("%s: %s", what, hipGetErrorString(q)); return MRGFE_ERR_HIP; }
        }
    }
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e)); return MRGFE_ERR_HIP; }
    if (done()) return MRGFE_OK;
    set_error("%s: the kernel did not report", what);
    return MRGFE_ERR_HIP;
}

// persistent host worker threads for the per-pair controller steps of large batches (a 6x6 SVD solve and the line-search
// bookkeeping per pair and round: ~3 us each, which adds up to the kernel time of a round once a batch has >100 pairs)
void host_parallel_for(int n, int min_serial, const std::function<void(int, int)>& body);
// workers spin between host_parallel_for calls while hot (set around the rounds of a large alignment batch), sleep otherwise
void host_parallel_hot(bool hot);

// bump allocator over a few large device chunks; pointers stay valid until reset()
struct Arena {
    struct Chunk { void* p; size_t cap; size_t used; };
    std::vector<Chunk> chunks;
    size_t chunk_bytes = size_t(64) << 20;
    int  alloc(size_t bytes, void** out);
    void reset();    // keep chunks, forget allocations
    void release();  // free chunks
};

}  // namespace mrgfe

// every extern "C" entry point that touches the GPU takes the context lock: handles of one context may be used from
// several threads (the reference's odometry and loop-closure registrations live in different threads of one process)
#define MRGFE_LOCK(ctxptr) std::lock_guard<std::recursive_mutex> _mrgfe_lock((ctxptr)->mu)

struct mrgfe_ctx;
namespace mrgfe {
class NnGrid;
void ctx_tmp_grid_free(mrgfe_ctx* ctx);  // nn_grid.hip
}

namespace mrgfe {
// what the last nn_fitness_batch on a context did (getFitnessScore passes): HIP-event times of its passes and the walk's counters
struct FitStats {
    double   ms_block = 0, ms_sweep = 0, ms_far = 0;   // block pass / seed + sweep / pyramid walk of the unseeded rest
    uint64_t queries = 0, queued = 0, queued_far = 0;  // all queries / not settled by their 3x3x3 block / not seeded within three blocks
    uint64_t words = 0, tested = 0, cells = 0, points = 0;  // seed + sweep: occupancy words fetched, boxes tested against the sphere, cells opened, points measured (MRGFE_FIT_STATS=1)
    uint64_t calls = 0;
    void add(const FitStats& o)
    {
        ms_block += o.ms_block; ms_sweep += o.ms_sweep; ms_far += o.ms_far;
        queries += o.queries; queued += o.queued; queued_far += o.queued_far;
        words += o.words; tested += o.tested; cells += o.cells; points += o.points;
        calls += 1;
    }
};
// the last k-NN launch on a context (NnGrid::knn_device: GICP covariances, StatisticalOutlierRemoval, mrgfe_knn)
struct KnnStats {
    hipEvent_t ev[2] = {nullptr, nullptr};
    DevBuf     counter;      // candidates measured (diagnostic counters on)
    uint64_t   queries = 0, launches = 0;
    int        k = 0;
    bool       counted = false;
};
}  // namespace mrgfe

struct mrgfe_ctx {
    int          device = 0;
    hipStream_t  stream = nullptr;
    hipEvent_t   ev0 = nullptr, ev1 = nullptr;  // timing of the dominant kernel on `stream`
    hipEvent_t   ev_mode[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // per NDT kernel variant
    mrgfe::DevBuf scratch[14];                  // named by the algorithms that use them
    mrgfe::DevBuf sort_chunks;                  // radix_sort_pairs: digit counts per chunk of 64 tiles (sorts of thousands of tiles: the map cloud)
    mrgfe::PinBuf pin[4];
    mrgfe::PinBuf up_pin[2];                    // upload_cloud staging ring: the host packs cloud k + 1 while cloud k is on the wire
    mrgfe::DevBuf up_raw, up_out;               // raw strided records waiting for the device gather / packed result of mrgfe_ingest_pointcloud2
    hipEvent_t   up_ev[2] = {nullptr, nullptr};
    bool         up_busy[2] = {false, false};
    int          up_next = 0;
    bool         dma_from_caller = false;     // a zero-copy upload was queued since the last wait for the stream (drain_caller_dma on error paths)
    bool         zero_copy_uploads = false;   // mrgfe_ctx_set_zero_copy_uploads: clouds in page-locked host memory go up by DMA from the caller's buffer
    hipEvent_t   ev_fit[4] = {nullptr, nullptr, nullptr, nullptr};  // around the passes of nn_fitness_batch
    int          priority = 0;                  // > 0: streams at the device's highest priority, < 0: at its lowest (throughput work beside latency-critical rounds)
    std::vector<uint32_t> cu_mask;              // non-empty: every stream of this context is confined to these compute units (mrgfe_ctx_create_reserving)
    int          make_stream(hipStream_t* st) const;  // a further stream of this context: same compute-unit mask
    hipStream_t  side = nullptr;                // second stream of nn_fitness_batch: the pyramid walk of the unseeded queries beside the sweep
    hipEvent_t   ev_side[4] = {nullptr, nullptr, nullptr, nullptr};  // fork, start and end of the side work, join
    mrgfe::FitStats fit_stats;                  // of the last nn_fitness_batch on this context
    mrgfe::KnnStats knn_stats;                  // of the last k-NN launch on this context
    // descriptor staging ring: small host tables (job records, offsets, slice tables) whose owner does not outlive the call that
    // enqueues their copy go through one of these pinned slots; a slot is reused only after the event behind its copy has passed
    static constexpr int kStageSlots = 8;
    mrgfe::PinBuf stage_pin[kStageSlots];
    hipEvent_t   stage_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool         stage_busy[kStageSlots] = {false, false, false, false, false, false, false, false};
    int          stage_next = 0;
    int          stage_h2d(void* d_dst, const void* src, size_t bytes, hipStream_t st);  // stream-ordered copy; `src` is free on return
    mrgfe::DevBuf pf_buf[2], pf_state;          // prefilter chain: ping-pong clouds and the device-resident state record (filters.hip)
    mrgfe::PinBuf pf_status;                    // ... and the few words the host reads at the chain's single wait
    void*        pf_grid = nullptr;             // NnDeviceDrivenGrid of the radius filter (nn_grid.h), created on first use
    // what the last mrgfe_prefilter_device of the device-driven chain left in the caller's buffer: the cloud, its size and a box that ENCLOSES it (the box of the
    // voxel centroids before the outlier filter, read with the chain's status words) — mrgfe_reg_set_source_from_prefilter builds the source's search grid on it
    const void*  pf_out_ptr = nullptr;
    size_t       pf_out_n = 0;
    float        pf_out_box[6] = {0, 0, 0, 0, 0, 0};  // min xyz, max xyz
    bool         pf_out_valid = false;
    int          cu_count = 256;
    mrgfe::NnGrid* tmp_grid = nullptr;          // reusable exact-NN grid of the stateless filter / fitness calls (nn_grid.hip)
    std::recursive_mutex mu;                    // serialises API calls that share this context's stream / workspaces
    int          bind();                        // hipSetDevice(device)
};

namespace mrgfe {
// copy a host cloud into packed float4 device memory via the pinned staging ring; `layout` is the stride_bytes argument of the C
// ABI (16, or MRGFE_LAYOUT(stride, xyz offset, intensity offset): such records are gathered on the device)
int upload_cloud(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t layout, void* d_dst, int pin_slot = 0);
void drain_caller_dma(mrgfe_ctx* ctx);
// roctx ranges around the phases of a call (set_target / rounds / fitness / gather): visible to rocprofv3 --marker-trace and to any roctx consumer; libroctx64 is
// found through dlopen, and the range is a no-op when it is not there
struct TraceRange {
    explicit TraceRange(const char* name);
    ~TraceRange();
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    bool on;
};  // error paths: wait for zero-copy uploads still reading the caller's page-locked buffers
int decode_layout(size_t layout, uint32_t* stride, uint32_t* xyz_off, int32_t* intensity_off);
// a helper context of `parent` (builder threads of a batch, GICP lanes): same device, same compute-unit mask
int ctx_create_like(const mrgfe_ctx* parent, mrgfe_ctx** out, int priority = 0);  // priority: see mrgfe_ctx::priority

}  // namespace mrgfe

// csrc/api.cpp — the extern "C" surface declared in include/mrgfe.h.  Thin: argument checks, column-major <-> row-major
// conversion, and dispatch into the engines.  Never throws; failures set the thread-local message.
#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <chrono>
#include <memory>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "cellsort.h"
#include "common.h"
#include "glibc_exp.h"
#include "filters.h"
#include "mapcloud.h"
#include "gicp_engine.h"
#include "ndt_derivatives.h"
#include "ndt_engine.h"
#include "nn_grid.h"

using namespace mrgfe;

namespace {

void col2row(const float in[16], float out[16]) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = in[c * 4 + r]; }
void row2col(const float in[16], float out[16]) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[c * 4 + r] = in[r * 4 + c]; }

// point counts the library accepts: the kernels index points with 32-bit words (and a count beyond 2^31 is a caller's bug, not a cloud)
constexpr size_t kMaxPoints = 0x7fffffffu;
int check_count(size_t n, const char* fn)
{
    if (n > kMaxPoints) { set_error("%s: %zu points: more than 2^31 - 1", fn, n); return MRGFE_ERR_INVALID; }
    return MRGFE_OK;
}

bool is_ndt(int method) { return method == MRGFE_NDT_HIP || method == MRGFE_PCL_NDT_HIP; }

NdtParams ndt_params_from(const mrgfe_reg_params& p)
{
    NdtParams n;
    n.resolution = static_cast<float>(p.resolution);
    n.step_size = p.step_size;
    n.outlier_ratio = p.outlier_ratio;
    n.trans_eps = p.transformation_epsilon;
    n.max_iterations = p.maximum_iterations;
    n.search = p.nn_search_method;
    if (p.method == MRGFE_PCL_NDT_HIP) {  // pcl::NormalDistributionsTransform has one neighbourhood: target_cells_.radiusSearch(point, resolution_)
        n.formulation = 1;
        n.search = MRGFE_KDTREE;
    }
    return n;
}
GicpParams gicp_params_from(const mrgfe_reg_params& p)
{
    GicpParams g;
    g.k_correspondences = p.correspondence_randomness;
    g.max_corr_dist = p.max_correspondence_distance;
    g.trans_eps = p.transformation_epsilon;
    g.rot_eps = p.rotation_epsilon;
    g.max_iterations = p.maximum_iterations;
    g.variant = p.method == MRGFE_SMALL_GICP_HIP ? 1 : p.method == MRGFE_VGICP_HIP ? 2 : p.method == MRGFE_ICP_HIP ? 3 : (p.method == MRGFE_PCL_GICP_HIP || p.method == MRGFE_PCL_GICP_OMP_HIP) ? 4 : 0;
    g.max_inner_iterations = p.max_optimizer_iterations;
    g.pcl_whole_gradient_norm = p.method == MRGFE_PCL_GICP_OMP_HIP;
    // pclomp::GICP never sees reg_num_threads (registrations.cpp:104-114 does not call setNumThreads): its sums are those of omp_get_max_threads() threads
    // of the reference's host.  Here: num_threads of the params when given, 8 (the YAML's reg_num_threads) otherwise; 2..16 chains (check_params refuses more).
    // ONE thread is the serial chain over the correspondences (one partial, 0 + p_0 = p_0): pcl::GICP's order (it used to fall through to the tree)
    const int omp_threads = p.method == MRGFE_PCL_GICP_OMP_HIP ? (p.num_threads > 0 ? p.num_threads : 8) : 0;
    g.pcl_reference_order_sums = p.method == MRGFE_PCL_GICP_HIP || omp_threads == 1;
    g.pcl_omp_sum_threads = omp_threads > 1 ? std::min(16, omp_threads) : 0;
    g.use_reciprocal = p.method == MRGFE_ICP_HIP && p.use_reciprocal_correspondences != 0;
    g.voxel_resolution = p.resolution;
    return g;
}

int check_params(const mrgfe_reg_params* p)
{
    if (!p) { set_error("NULL params"); return MRGFE_ERR_INVALID; }
    if (p->method == MRGFE_PCL_GICP_OMP_HIP && p->num_threads > 16) {
        set_error("PCL_GICP_OMP_HIP adds its cost terms as num_threads OpenMP threads would (static chunks, partials in thread order): 1..16 threads are supported, got %d", p->num_threads);
        return MRGFE_ERR_INVALID;
    }
    if (p->method < MRGFE_NDT_HIP || p->method > MRGFE_PCL_NDT_HIP) { set_error("unknown registration method %d", p->method); return MRGFE_ERR_INVALID; }
    if ((p->method == MRGFE_PCL_GICP_HIP || p->method == MRGFE_PCL_GICP_OMP_HIP) && p->max_optimizer_iterations < 1) { set_error("max_optimizer_iterations must be >= 1"); return MRGFE_ERR_INVALID; }
    if (is_ndt(p->method)) {
        if (!(p->resolution > 0)) { set_error("resolution must be > 0"); return MRGFE_ERR_INVALID; }
        if (p->method == MRGFE_NDT_HIP && (p->nn_search_method < 0 || p->nn_search_method > 3)) { set_error("unknown nn_search_method %d", p->nn_search_method); return MRGFE_ERR_INVALID; }
    } else {
        if (p->method != MRGFE_ICP_HIP && (p->correspondence_randomness < 4 || p->correspondence_randomness > 64)) { set_error("correspondence_randomness must be in [4, 64]"); return MRGFE_ERR_INVALID; }
        if (p->method == MRGFE_VGICP_HIP && !(p->resolution > 0)) { set_error("resolution must be > 0"); return MRGFE_ERR_INVALID; }
    }
    return MRGFE_OK;
}

}  // namespace

struct mrgfe_reg {
    mrgfe_ctx*       ctx = nullptr;
    mrgfe_reg_params params;
    NdtEngine*       ndt = nullptr;
    GicpEngine*      gicp = nullptr;
    DevBuf           tgt, src;            // owned copies of host-supplied clouds
    const void*      d_tgt = nullptr;     // current clouds (owned buffer or caller's device memory)
    const void*      d_src = nullptr;
    size_t           n_tgt = 0, n_src = 0;
    bool             has_target = false, has_source = false, aligned_once = false;
    int              target_status = MRGFE_ERR_STATE;
    NnGrid           nn;                  // exact 1-NN structure over the target (getFitnessScore / nearestKSearch)
    bool             nn_valid = false;
    float            final_rm[16];        // row-major
    bool             converged = false;
    int              iterations = 0, evaluations = 0;
    double           trans_probability = 0;
    double           hessian[36];
    double           mean_neighbours = 0;
};

struct mrgfe_batch {
    mrgfe_ctx*       ctx = nullptr;
    mrgfe_reg_params params;
    NdtEngine*       ndt = nullptr;   // holds the clouds, targets and pairs of the batch for both methods; aligns them for NDT_HIP
    std::vector<GicpEngine*> gicp;    // GICP_HIP: one engine per target (its covariances and correspondence grid are computed once)
    std::vector<float>  gicp_final;   // GICP_HIP: row-major final transformation of every pair
    GicpBatch*          gicp_batch = nullptr;
    std::vector<GicpBatchPair> gicp_pairs;  // per-pair device buffers, kept between align calls
    std::vector<NnGrid> fit_grids;  // getFitnessScore grids, one per target; device buffers kept between align calls
    std::vector<std::unique_ptr<NnGridSet>> fit_sets;  // ... which are views into these when the grids were built a chunk of targets at a time
    mrgfe_ctx* early_ctx = nullptr;         // lowest-priority context of the early fitness pass (mrgfe_batch_align)
    std::vector<mrgfe_ctx*> fit_ctxs;       // helper contexts (own stream and workspaces each): the grids are built on them by extra host
                                            // thread while the alignment rounds run on the batch's context
    // keyframe store (mrgfe_batch_add_pair_keyed): packed clouds and GICP covariances by caller-chosen key, resident across clears
    struct Keyframe {
        DevBuf   cloud, cov;
        uint32_t n = 0;
        int      cov_k = 0;       // k_correspondences the covariances were computed with; 0: none yet
        uint64_t last_epoch = 0;  // batch epoch (number of clears) of the last use
        uint64_t last_tick = 0;
        size_t   bytes() const { return cloud.cap + cov.cap; }
    };
    std::unordered_map<uint64_t, Keyframe*> store;
    std::vector<uint64_t> pair_key;  // per pair; 0: not from the store
    FitStats   fit_total;             // getFitnessScore passes of the last align(), all waves added up
    std::unique_ptr<NdtSnapshotPort> port;  // early fitness passes (mrgfe_batch_align)
    hipEvent_t uploads_done = nullptr;  // recorded on ctx->stream before helper streams read the batch's clouds (upload_cloud is stream-ordered only)
    uint64_t epoch = 1, tick = 0;
    size_t   store_cap = size_t(16384) << 20;
    // mrgfe_batch_timing: the reference's per-candidate time (loop_detector.cpp:22-34): from the clear that starts queueing a batch to its records
    std::chrono::steady_clock::time_point t_queue;
    bool     t_queue_set = false;
    double   last_us = 0.0, total_us = 0.0;
    int64_t  last_pairs = 0, total_pairs = 0;
    // mrgfe_batch_align_async / mrgfe_batch_wait: a worker thread of the batch's own runs mrgfe_batch_align while the caller queues the next batch
    // on another object.  It takes the context lock BEFORE the async call returns, so every other call on this batch simply waits for the align.
    struct Async {
        std::thread th;
        std::mutex  mu;
        std::condition_variable cv;
        int    state = 0;  // 0 idle, 1 posted, 2 running (context lock held), 3 finished and not yet waited for
        bool   quit = false;
        double fitness_max_range = -1.0;
        mrgfe_pair_result* results = nullptr;
        int    status = MRGFE_OK;
        std::string error;
    };
    std::unique_ptr<Async> async;
};

// drop least recently used keyframes that the current batch does not reference until `need` more bytes fit
static void store_make_room(mrgfe_batch* b, size_t need)
{
    size_t total = 0;
    for (auto& kv : b->store) total += kv.second->bytes();
    while (total + need > b->store_cap) {
        uint64_t victim = 0, best = ~uint64_t(0);
        for (auto& kv : b->store)
            if (kv.second->last_epoch < b->epoch && kv.second->last_tick < best) { best = kv.second->last_tick; victim = kv.first; }
        if (!victim) return;  // everything left is in use: the store grows past its cap for this batch
        auto it = b->store.find(victim);
        total -= it->second->bytes();
        it->second->cloud.release();
        it->second->cov.release();
        delete it->second;
        b->store.erase(it);
    }
}

extern "C" {

void mrgfe_reg_default_params(int method, mrgfe_reg_params* out)
{
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    out->method = method;
    out->num_threads = 0;                       // registrations.cpp:35
    out->transformation_epsilon = 0.01;         // :36
    out->maximum_iterations = 64;               // :37
    out->max_correspondence_distance = 2.0;     // :38
    out->max_optimizer_iterations = 20;         // :39
    out->use_reciprocal_correspondences = 0;    // :40
    out->correspondence_randomness = 20;        // :41
    out->resolution = 1.0;                      // :42
    out->nn_search_method = MRGFE_DIRECT7;      // :43
    out->step_size = 0.1;
    out->outlier_ratio = 0.55;
    out->rotation_epsilon = 2e-3;
}

int mrgfe_reg_create(mrgfe_ctx* ctx, const mrgfe_reg_params* params, mrgfe_reg** out)
{
    if (!ctx || !out) { set_error("mrgfe_reg_create: NULL argument"); return MRGFE_ERR_INVALID; }
    *out = nullptr;
    MRGFE_TRY(check_params(params));
    mrgfe_reg* r = new (std::nothrow) mrgfe_reg();
    if (!r) { set_error("out of host memory"); return MRGFE_ERR_INVALID; }
    r->ctx = ctx;
    r->params = *params;
    if (is_ndt(params->method)) {
        r->ndt = new NdtEngine(ctx, ndt_params_from(*params));
        if (const char* e = std::getenv("MRGFE_FORCE_HASH")) r->ndt->set_force_hash(e[0] == '1');
    } else {
        r->gicp = new GicpEngine(ctx, gicp_params_from(*params));
    }
    for (int i = 0; i < 16; ++i) r->final_rm[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    for (int i = 0; i < 36; ++i) r->hessian[i] = 0;
    *out = r;
    return MRGFE_OK;
}

void mrgfe_reg_destroy(mrgfe_reg* reg)
{
    if (!reg) return;
    MRGFE_LOCK(reg->ctx);
    (void)hipSetDevice(reg->ctx->device);
    delete reg->ndt;
    delete reg->gicp;
    reg->nn.release();
    reg->tgt.release();
    reg->src.release();
    delete reg;
}

static int reg_target_changed(mrgfe_reg* reg)
{
    reg->has_target = true;
    reg->nn_valid = false;
    if (reg->ndt) {
        reg->ndt->clear();
        int ti = reg->ndt->add_target_device(reg->d_tgt, reg->n_tgt);
        if (ti < 0) return ti;
        MRGFE_TRY(reg->ndt->build_targets());
        reg->target_status = reg->ndt->target(0).status;
        if (reg->target_status == MRGFE_ERR_OVERFLOW) set_error("[NDT_HIP::setInputTarget] Leaf size is too small for the input dataset. Integer indices would overflow.");
        if (reg->target_status == MRGFE_ERR_EMPTY) { set_error("setInputTarget: cloud has no finite point"); }
        return reg->target_status;
    }
    MRGFE_TRY(reg->gicp->set_target(reg->d_tgt, reg->n_tgt));
    reg->target_status = MRGFE_OK;
    return MRGFE_OK;
}

int mrgfe_reg_set_target(mrgfe_reg* reg, const float* xyzi, size_t n, size_t stride_bytes)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_set_target"));
    if (!reg || (n && !xyzi)) { set_error("mrgfe_reg_set_target: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    MRGFE_TRY(reg->tgt.ensure(std::max<size_t>(n, 1) * 16));
    MRGFE_TRY(upload_cloud(reg->ctx, xyzi, n, stride_bytes, reg->tgt.p));
    reg->d_tgt = reg->tgt.p;
    reg->n_tgt = n;
    TraceRange tr("mrgfe_reg_set_target");
    return reg_target_changed(reg);
}

int mrgfe_reg_set_target_device(mrgfe_reg* reg, const void* d_xyzi, size_t n)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_set_target_device"));
    if (!reg || (n && !d_xyzi)) { set_error("mrgfe_reg_set_target_device: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    reg->d_tgt = d_xyzi;
    reg->n_tgt = n;
    return reg_target_changed(reg);
}

int mrgfe_reg_set_source(mrgfe_reg* reg, const float* xyzi, size_t n, size_t stride_bytes)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_set_source"));
    if (!reg || (n && !xyzi)) { set_error("mrgfe_reg_set_source: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    MRGFE_TRY(reg->src.ensure(std::max<size_t>(n, 1) * 16));
    MRGFE_TRY(upload_cloud(reg->ctx, xyzi, n, stride_bytes, reg->src.p));
    reg->d_src = reg->src.p;
    reg->n_src = n;
    reg->has_source = true;
    if (reg->gicp) MRGFE_TRY(reg->gicp->set_source(reg->d_src, reg->n_src));
    return MRGFE_OK;
}

int mrgfe_reg_set_source_device(mrgfe_reg* reg, const void* d_xyzi, size_t n)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_set_source_device"));
    if (!reg || (n && !d_xyzi)) { set_error("mrgfe_reg_set_source_device: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    reg->d_src = d_xyzi;
    reg->n_src = n;
    reg->has_source = true;
    if (reg->gicp) MRGFE_TRY(reg->gicp->set_source(reg->d_src, reg->n_src));
    return MRGFE_OK;
}

int mrgfe_reg_set_source_from_prefilter(mrgfe_reg* reg, const void* d_xyzi, size_t n)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_set_source_from_prefilter"));
    if (!reg || (n && !d_xyzi)) { set_error("mrgfe_reg_set_source_from_prefilter: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    reg->d_src = d_xyzi;
    reg->n_src = n;
    reg->has_source = true;
    const mrgfe_ctx* c = reg->ctx;
    const bool boxed = c->pf_out_valid && c->pf_out_ptr == d_xyzi && c->pf_out_n == n && n > 0;
    if (reg->gicp) MRGFE_TRY(reg->gicp->set_source(reg->d_src, reg->n_src, boxed ? c->pf_out_box : nullptr));
    return MRGFE_OK;
}

int mrgfe_reg_source_becomes_target(mrgfe_reg* reg)
{
    if (!reg) { set_error("mrgfe_reg_source_becomes_target: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!reg->has_source) { set_error("mrgfe_reg_source_becomes_target: setInputSource first"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    TraceRange tr("mrgfe_reg_source_becomes_target");
    // a cloud the library uploaded lives in reg->src: that buffer becomes the target's, the old target's takes the next source
    if (reg->d_src == reg->src.p && reg->src.p != nullptr) std::swap(reg->src, reg->tgt);
    reg->d_tgt = reg->d_src;
    reg->n_tgt = reg->n_src;
    if (reg->gicp) {
        reg->has_target = true;
        reg->nn_valid = false;
        MRGFE_TRY(reg->gicp->source_becomes_target());
        reg->target_status = MRGFE_OK;
        return MRGFE_OK;
    }
    return reg_target_changed(reg);  // NDT: the voxel grid of the new target (a source has nothing to hand over)
}

int mrgfe_reg_align(mrgfe_reg* reg, const float guess[16], float* aligned_xyzi)
{
    if (!reg || !guess) { set_error("mrgfe_reg_align: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!reg->has_target || !reg->has_source) { set_error("align: setInputTarget / setInputSource first"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    TraceRange tr("mrgfe_reg_align");
    float g[16];
    col2row(guess, g);
    if (reg->ndt) {
        NdtEngine& e = *reg->ndt;
        e.clear_pairs();
        int pi = e.add_pair_device(0, reg->d_src, reg->n_src, g);
        if (pi < 0) return pi;
        MRGFE_TRY(e.align_all());
        const NdtController& c = e.pair(0).ctl;
        std::memcpy(reg->final_rm, c.final_transformation(), sizeof(reg->final_rm));
        reg->converged = c.converged();
        reg->iterations = c.iterations();
        reg->evaluations = c.evaluations();
        reg->trans_probability = c.trans_probability();
        std::memcpy(reg->hessian, c.hessian(), sizeof(reg->hessian));
        reg->mean_neighbours = c.evaluations() ? c.neighbours_sum() / c.evaluations() : 0.0;
        if (aligned_xyzi) MRGFE_TRY(e.aligned_cloud(0, aligned_xyzi));
    } else {
        GicpEngine& e = *reg->gicp;
        MRGFE_TRY(e.align(g));
        std::memcpy(reg->final_rm, e.final_transformation(), sizeof(reg->final_rm));
        reg->converged = e.converged();
        reg->iterations = e.iterations();
        reg->evaluations = e.evaluations();
        reg->trans_probability = 0;
        std::memcpy(reg->hessian, e.hessian(), sizeof(reg->hessian));
        if (aligned_xyzi) MRGFE_TRY(e.aligned_cloud(aligned_xyzi));
    }
    reg->aligned_once = true;
    return MRGFE_OK;
}

int mrgfe_reg_has_converged(const mrgfe_reg* reg) { return reg && reg->converged ? 1 : 0; }

int mrgfe_reg_final_transformation(const mrgfe_reg* reg, float out[16])
{
    if (!reg || !out) { set_error("mrgfe_reg_final_transformation: NULL argument"); return MRGFE_ERR_INVALID; }
    row2col(reg->final_rm, out);
    return MRGFE_OK;
}

static int reg_ensure_nn(mrgfe_reg* reg)
{
    if (!reg->has_target) { set_error("no target set"); return MRGFE_ERR_STATE; }
    if (!reg->nn_valid) {
        MRGFE_TRY(reg->nn.build(reg->ctx, static_cast<const float4*>(reg->d_tgt), reg->n_tgt, 1.0f, NnGrid::kCrowding1nn, 1));
        reg->nn_valid = true;
    }
    return MRGFE_OK;
}

int mrgfe_reg_fitness(mrgfe_reg* reg, double max_range, double* out)
{
    TraceRange tr("mrgfe_reg_fitness");
    if (!reg || !out) { set_error("mrgfe_reg_fitness: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!reg->has_target || !reg->has_source) { set_error("getFitnessScore: target / source not set"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    if (reg->n_tgt == 0 || reg->n_src == 0) { *out = DBL_MAX; return MRGFE_OK; }
    MRGFE_TRY(reg_ensure_nn(reg));
    return reg->nn.fitness(reg->ctx, static_cast<const float4*>(reg->d_src), reg->n_src, reg->final_rm, max_range, out);
}

int mrgfe_reg_nn1_target(mrgfe_reg* reg, const float* q, size_t n, size_t stride_bytes, int32_t* idx, float* sqd)
{
    MRGFE_TRY(check_count(n, "mrgfe_reg_nn1_target"));
    if (!reg || (n && (!q || !idx || !sqd))) { set_error("mrgfe_reg_nn1_target: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    MRGFE_TRY(reg_ensure_nn(reg));
    return reg->nn.nearest_host(reg->ctx, q, n, stride_bytes, idx, sqd);
}

int    mrgfe_reg_iterations(const mrgfe_reg* reg) { return reg ? reg->iterations : 0; }
int    mrgfe_reg_evaluations(const mrgfe_reg* reg) { return reg ? reg->evaluations : 0; }
double mrgfe_reg_trans_probability(const mrgfe_reg* reg) { return reg ? reg->trans_probability : 0.0; }
int    mrgfe_reg_hessian(const mrgfe_reg* reg, double out[36])
{
    if (!reg || !out) { set_error("mrgfe_reg_hessian: NULL argument"); return MRGFE_ERR_INVALID; }
    std::memcpy(out, reg->hessian, sizeof(reg->hessian));
    return MRGFE_OK;
}

// ---- NDT internals --------------------------------------------------------------------------------------------
int mrgfe_ndt_evaluate(mrgfe_reg* reg, const float T[16], const double p[6], int mode, double* score, double grad[6], double hess[36])
{
    if (!reg || !reg->ndt || !T || !p || !score || !grad || !hess) { set_error("mrgfe_ndt_evaluate: needs an NDT registration and non-NULL arguments"); return MRGFE_ERR_INVALID; }
    if (!reg->has_target || !reg->has_source) { set_error("evaluate: target / source not set"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    float Tr[16];
    col2row(T, Tr);
    NdtEngine& e = *reg->ndt;
    e.clear_pairs();
    float ident[16];
    for (int i = 0; i < 16; ++i) ident[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    int pi = e.add_pair_device(0, reg->d_src, reg->n_src, ident);
    if (pi < 0) return pi;
    return e.evaluate(0, Tr, p, mode, score, grad, hess);
}

int mrgfe_knn(mrgfe_ctx* ctx, const float* cloud, size_t n, const float* query, size_t nq, size_t stride, int k, int32_t* idx, float* sqd)
{
    MRGFE_TRY(check_count(n, "mrgfe_knn"));
    MRGFE_TRY(check_count(nq, "mrgfe_knn"));
    if (!ctx || (n && !cloud) || (nq && (!query || !idx || !sqd))) { set_error("mrgfe_knn: NULL argument"); return MRGFE_ERR_INVALID; }
    if (k < 1 || k > 64) { set_error("mrgfe_knn: k must be in [1, 64]"); return MRGFE_ERR_INVALID; }
    if (nq == 0) return MRGFE_OK;
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    DevBuf dc, dq, di, dd;
    NnGrid grid;
    int rc = dc.ensure(std::max<size_t>(n, 1) * 16);
    if (rc == MRGFE_OK) rc = dq.ensure(nq * 16);
    if (rc == MRGFE_OK) rc = di.ensure(nq * k * 4);
    if (rc == MRGFE_OK) rc = dd.ensure(nq * k * 4);
    if (rc == MRGFE_OK && n) rc = upload_cloud(ctx, cloud, n, stride, dc.p);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, query, nq, stride, dq.p);
    if (rc == MRGFE_OK) rc = grid.build(ctx, dc.as<float4>(), n, 1.0f, NnGrid::kCrowdingKnn);
    if (rc == MRGFE_OK) rc = grid.knn_device(ctx, dq.as<float4>(), nq, k, di.as<int32_t>(), dd.as<float>());
    if (rc == MRGFE_OK && (hipMemcpyAsync(idx, di.p, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                           hipMemcpyAsync(sqd, dd.p, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        set_error("mrgfe_knn: device to host copy failed");
        rc = MRGFE_ERR_HIP;
    }
    grid.release();
    dc.release(); dq.release(); di.release(); dd.release();
    return rc;
}

int mrgfe_dbg_set_gicp_corr_passes(int mode) { return gicp_set_corr_passes(mode); }

int mrgfe_dbg_grid_set_query(mrgfe_ctx* ctx, const float* const* clouds, const size_t* n, int count, const float* query, size_t nq, int k, int rounds, int32_t* idx, float* sqd)
{
    if (!ctx || count < 1 || !clouds || !n || !query || !idx || !sqd || nq == 0) { set_error("mrgfe_dbg_grid_set_query: bad argument"); return MRGFE_ERR_INVALID; }
    if (k < 1 || k > 64) { set_error("mrgfe_dbg_grid_set_query: k must be in [1, 64]"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    std::vector<DevBuf>        dc(count);
    std::vector<NnGrid>        grids(count);
    std::vector<NnGrid*>       gp(count);
    std::vector<const float4*> cp(count);
    std::vector<uint32_t>      nn(count);
    DevBuf     dq, di, dd;
    NnGridSet  set;
    int rc = dq.ensure(nq * 16);
    if (rc == MRGFE_OK) rc = di.ensure(nq * k * 4);
    if (rc == MRGFE_OK) rc = dd.ensure(nq * k * 4);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, query, nq, 16, dq.p);
    for (int m = 0; m < count && rc == MRGFE_OK; ++m) {
        rc = dc[m].ensure(std::max<size_t>(n[m], 1) * 16);
        if (rc == MRGFE_OK && n[m]) rc = upload_cloud(ctx, clouds[m], n[m], 16, dc[m].p);
        cp[m] = dc[m].as<float4>();
        nn[m] = static_cast<uint32_t>(n[m]);
        gp[m] = &grids[m];
    }
    // (built `rounds` times: the second and later builds start from the first one's cell edges)
    for (int r = 0; r < std::max(1, rounds) && rc == MRGFE_OK; ++r)
        rc = k == 1 ? set.build(ctx, cp.data(), nn.data(), count, 1.0f, NnGrid::kCrowding1nn, 1, gp.data()) : set.build(ctx, cp.data(), nn.data(), count, 1.0f, NnGrid::kCrowdingKnn, kNnMaxLevels, gp.data());
    for (int m = 0; m < count && rc == MRGFE_OK; ++m) {
        rc = k == 1 ? grids[m].nearest_device(ctx, dq.as<float4>(), nq, nullptr, di.as<int32_t>(), dd.as<float>()) : grids[m].knn_device(ctx, dq.as<float4>(), nq, k, di.as<int32_t>(), dd.as<float>());
        if (rc == MRGFE_OK && (hipMemcpyAsync(idx + size_t(m) * nq * k, di.p, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                               hipMemcpyAsync(sqd + size_t(m) * nq * k, dd.p, nq * k * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
            set_error("mrgfe_dbg_grid_set_query: device to host copy failed");
            rc = MRGFE_ERR_HIP;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MRGFE_OK) rc = MRGFE_ERR_HIP;
    set.release();
    for (auto& b : dc) b.release();
    dq.release(); di.release(); dd.release();
    return rc;
}

int mrgfe_gicp_linearize(mrgfe_reg* reg, const double T[16], double H[36], double b[6], double* sum_errors, int* n_correspondences)
{
    if (!reg || !reg->gicp || !T || !H || !b || !sum_errors || !n_correspondences) { set_error("mrgfe_gicp_linearize: needs a GICP registration and non-NULL arguments"); return MRGFE_ERR_INVALID; }
    if (reg->params.method == MRGFE_ICP_HIP) { set_error("mrgfe_gicp_linearize: ICP_HIP has no linearised cost"); return MRGFE_ERR_INVALID; }
    if (reg->params.method == MRGFE_PCL_GICP_HIP || reg->params.method == MRGFE_PCL_GICP_OMP_HIP) { set_error("mrgfe_gicp_linearize: PCL_GICP_HIP minimises with BFGS (mrgfe_pclgicp_evaluate)"); return MRGFE_ERR_INVALID; }
    if (!reg->has_target || !reg->has_source) { set_error("linearize: target / source not set"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    double Tr[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Tr[r * 4 + c] = T[c * 4 + r];
    return reg->gicp->linearize(Tr, H, b, sum_errors, n_correspondences);
}

int mrgfe_pclgicp_evaluate(mrgfe_reg* reg, const float T[16], const double x[6], double* f, double grad[6], int* n_correspondences)
{
    if (!reg || !reg->gicp || !T || !x || !f || !grad || !n_correspondences) { set_error("mrgfe_pclgicp_evaluate: NULL argument"); return MRGFE_ERR_INVALID; }
    if (reg->params.method != MRGFE_PCL_GICP_HIP && reg->params.method != MRGFE_PCL_GICP_OMP_HIP) { set_error("mrgfe_pclgicp_evaluate: needs a PCL_GICP_HIP registration"); return MRGFE_ERR_INVALID; }
    if (!reg->has_target || !reg->has_source) { set_error("evaluate: target / source not set"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    float Tr[16], eye[16];
    col2row(T, Tr);
    for (int i = 0; i < 16; ++i) eye[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    MRGFE_TRY(reg->gicp->covariances(0, nullptr));  // covariances, target grid and work buffers in place
    return reg->gicp->pcl_evaluate(Tr, eye, reg->gicp->source_points(), true, x, f, grad, n_correspondences);
}

int mrgfe_gicp_covariances(mrgfe_reg* reg, int which, double* cov9_per_point)
{
    if (!reg || !reg->gicp || !cov9_per_point || which < 0 || which > 1) { set_error("mrgfe_gicp_covariances: needs a GICP registration, which in {0, 1} and an output buffer"); return MRGFE_ERR_INVALID; }
    if (reg->params.method == MRGFE_ICP_HIP) { set_error("mrgfe_gicp_covariances: ICP_HIP has no covariances"); return MRGFE_ERR_INVALID; }
    if ((which == 0 && !reg->has_source) || (which == 1 && !reg->has_target)) { set_error("covariances: cloud not set"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    MRGFE_TRY(reg->ctx->bind());
    return reg->gicp->covariances(which, cov9_per_point);
}

int mrgfe_ndt_num_leaves(const mrgfe_reg* reg) { return (reg && reg->ndt && reg->ndt->n_targets() > 0) ? static_cast<int>(reg->ndt->target(0).n_leaves) : 0; }

int mrgfe_ndt_grid(const mrgfe_reg* reg, int32_t min_b[3], int32_t max_b[3], int32_t div_b[3])
{
    if (!reg || !reg->ndt || reg->ndt->n_targets() == 0) { set_error("mrgfe_ndt_grid: no NDT target"); return MRGFE_ERR_STATE; }
    const NdtTargetInfo& t = reg->ndt->target(0);
    for (int a = 0; a < 3; ++a) { min_b[a] = t.min_b[a]; max_b[a] = t.max_b[a]; div_b[a] = t.div_b[a]; }
    return MRGFE_OK;
}

int mrgfe_ndt_leaves(mrgfe_reg* reg, int32_t* keys, int32_t* nr_points, double* mean3, double* icov9)
{
    if (!reg || !reg->ndt || reg->ndt->n_targets() == 0) { set_error("mrgfe_ndt_leaves: no NDT target"); return MRGFE_ERR_STATE; }
    MRGFE_LOCK(reg->ctx);
    return reg->ndt->read_leaves(0, keys, nr_points, mean3, icov9);
}

double mrgfe_ndt_mean_neighbours(const mrgfe_reg* reg) { return reg ? reg->mean_neighbours : 0.0; }

int mrgfe_reg_kernel_stats(const mrgfe_reg* reg, int mode, double* ms, int64_t* launches, double* bytes)
{
    if (!reg) { set_error("NULL registration"); return MRGFE_ERR_INVALID; }
    double m = 0, b = 0;
    int64_t l = 0;
    if (reg->ndt) reg->ndt->kernel_stats(mode, &m, &l, &b);
    if (reg->gicp) { m = reg->gicp->kernel_ms; l = reg->gicp->kernel_launches; b = reg->gicp->kernel_alg_bytes; }
    if (ms) *ms = m;
    if (launches) *launches = l;
    if (bytes) *bytes = b;
    return MRGFE_OK;
}

// ---- prefilters -----------------------------------------------------------------------------------------------
int mrgfe_distance_filter(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double near_thresh, double far_thresh, float* out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_distance_filter"));
    if (!ctx || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_distance_filter: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    return filter_distance(ctx, xyzi, n, stride, near_thresh, far_thresh, out, out_n);
}
int mrgfe_approx_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, float* out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_approx_voxelgrid"));
    if (!ctx || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_approx_voxelgrid: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!(leaf > 0)) { set_error("mrgfe_approx_voxelgrid: leaf size must be > 0"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    return filter_approx_voxelgrid(ctx, xyzi, n, stride, leaf, out, out_n);
}
int mrgfe_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, float leaf, int min_pts, float* out, size_t* out_n, int* overflow)
{
    MRGFE_TRY(check_count(n, "mrgfe_voxelgrid"));
    if (!ctx || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_voxelgrid: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!(leaf > 0)) { set_error("mrgfe_voxelgrid: leaf size must be > 0"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    return filter_voxelgrid(ctx, xyzi, n, stride, leaf, min_pts, out, out_n, overflow);
}
int mrgfe_radius_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, double radius, int min_neighbors, float* out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_radius_outlier"));
    if (!ctx || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_radius_outlier: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!(radius > 0)) { set_error("mrgfe_radius_outlier: radius must be > 0"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    return filter_radius_outlier(ctx, xyzi, n, stride, radius, min_neighbors, out, out_n);
}
int mrgfe_statistical_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, int mean_k, double stddev_mul, float* out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_statistical_outlier"));
    if (!ctx || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_statistical_outlier: NULL argument"); return MRGFE_ERR_INVALID; }
    if (mean_k < 1 || mean_k > 63) { set_error("mrgfe_statistical_outlier: mean_k must be in [1, 63]"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    return filter_statistical_outlier(ctx, xyzi, n, stride, mean_k, stddev_mul, out, out_n);
}
void mrgfe_prefilter_default_params(mrgfe_prefilter_params* p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->enable_distance_filter = 1;           // config/mrg_slam.yaml:62-64
    p->distance_near_thresh = 0.1;
    p->distance_far_thresh = 35.0;
    p->downsample_method = 1;                // :48-50
    p->downsample_resolution = 0.1;
    p->downsample_min_points_per_voxel = 1;
    p->outlier_removal_method = 1;           // :53-59
    p->radius_radius = 0.5;
    p->radius_min_neighbors = 2;
    p->statistical_mean_k = 30;
    p->statistical_stddev = 1.2;
}
static int prefilter_impl(mrgfe_ctx* ctx, const mrgfe_prefilter_params* p, const float* xyzi, size_t n, size_t stride, void* out, size_t* out_n, bool on_device);
int mrgfe_prefilter(mrgfe_ctx* ctx, const mrgfe_prefilter_params* p, const float* xyzi, size_t n, size_t stride, float* out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_prefilter"));
    return prefilter_impl(ctx, p, xyzi, n, stride, out, out_n, false);
}
int mrgfe_prefilter_device(mrgfe_ctx* ctx, const mrgfe_prefilter_params* p, const float* xyzi, size_t n, size_t stride, void* d_out, size_t* out_n)
{
    MRGFE_TRY(check_count(n, "mrgfe_prefilter_device"));
    return prefilter_impl(ctx, p, xyzi, n, stride, d_out, out_n, true);
}
static int prefilter_impl(mrgfe_ctx* ctx, const mrgfe_prefilter_params* p, const float* xyzi, size_t n, size_t stride, void* out, size_t* out_n, bool on_device)
{
    if (!ctx || !p || !out_n || (n && (!xyzi || !out))) { set_error("mrgfe_prefilter: NULL argument"); return MRGFE_ERR_INVALID; }
    if (p->downsample_method < 0 || p->downsample_method > 2 || p->outlier_removal_method < 0 || p->outlier_removal_method > 2) { set_error("mrgfe_prefilter: unknown method"); return MRGFE_ERR_INVALID; }
    if (p->downsample_method >= 1 && !(p->downsample_resolution > 0)) { set_error("mrgfe_prefilter: downsample_resolution must be > 0"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    PrefilterChain ch;
    ch.distance = p->enable_distance_filter != 0;
    ch.near_t = p->distance_near_thresh;
    ch.far_t = p->distance_far_thresh;
    ch.voxelgrid = p->downsample_method == 1;
    ch.approx_voxelgrid = p->downsample_method == 2;
    ch.leaf = static_cast<float>(p->downsample_resolution);
    ch.min_pts = p->downsample_min_points_per_voxel;
    ch.outlier = p->outlier_removal_method;
    ch.radius = p->radius_radius;
    ch.radius_min_neighbors = p->radius_min_neighbors;
    ch.mean_k = p->statistical_mean_k;
    ch.stddev_mul = p->statistical_stddev;
    return filter_chain(ctx, ch, xyzi, n, stride, out, out_n, on_device);
}
int mrgfe_calc_fitness_score(mrgfe_ctx* ctx, const float* cloud1, size_t n1, const float* cloud2, size_t n2, size_t stride, const double relpose[16], double max_range, double* out)
{
    MRGFE_TRY(check_count(n1, "mrgfe_calc_fitness_score"));
    MRGFE_TRY(check_count(n2, "mrgfe_calc_fitness_score"));
    if (!ctx || !out || !relpose || (n1 && !cloud1) || (n2 && !cloud2)) { set_error("mrgfe_calc_fitness_score: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    if (n1 == 0 || n2 == 0) { *out = DBL_MAX; return MRGFE_OK; }
    DevBuf &d1 = ctx->scratch[10], &d2 = ctx->scratch[11];
    MRGFE_TRY(d1.ensure(n1 * 16));
    MRGFE_TRY(d2.ensure(n2 * 16));
    MRGFE_TRY(upload_cloud(ctx, cloud1, n1, stride, d1.p));
    MRGFE_TRY(upload_cloud(ctx, cloud2, n2, stride, d2.p));
    NnGrid& g = ctx_tmp_grid(ctx);
    int st = g.build(ctx, d1.as<float4>(), n1, 1.0f, NnGrid::kCrowding1nn, 1);
    if (st == MRGFE_OK) {
        float T[16];  // relpose.cast<float>(), row-major
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T[r * 4 + c] = static_cast<float>(relpose[c * 4 + r]);
        st = g.fitness(ctx, d2.as<float4>(), n2, T, max_range, out);
    }
    return st;
}

// ---- InformationMatrixCalculator (src/mrg_slam/information_matrix_calculator.cpp) ------------------------------------------
void mrgfe_inf_default_params(mrgfe_inf_params* p)
{
    if (!p) return;
    p->use_const_inf_matrix = 0;   // config/mrg_slam.yaml:216
    p->const_stddev_x = 0.5;       // :217
    p->const_stddev_q = 0.1;       // :218
    p->var_gain_a = 2.0;           // :219
    p->min_stddev_x = 0.1;         // :220
    p->max_stddev_x = 0.75;        // :221
    p->min_stddev_q = 0.05;        // :222
    p->max_stddev_q = 0.2;         // :223
    p->fitness_score_thresh = 1.25;  // :173
}
double mrgfe_inf_weight(double a, double max_x, double min_y, double max_y, double x)
{
    const double y = (1.0 - std::exp(-a * x)) / (1.0 - std::exp(-a * max_x));  // information_matrix_calculator.cpp:86
    return min_y + (max_y - min_y) * y;
}
int mrgfe_inf_matrix_from_fitness(const mrgfe_inf_params* p, double fitness_score, double inf[36])
{
    if (!p || !inf) { set_error("mrgfe_inf_matrix_from_fitness: NULL argument"); return MRGFE_ERR_INVALID; }
    double dx, dq;  // divisors of the two diagonal blocks
    if (p->use_const_inf_matrix) {  // :19-24 (divides by the standard deviation itself, as the reference does)
        dx = p->const_stddev_x;
        dq = p->const_stddev_q;
    } else {                        // :30-43
        const double min_var_x = std::pow(p->min_stddev_x, 2), max_var_x = std::pow(p->max_stddev_x, 2);
        const double min_var_q = std::pow(p->min_stddev_q, 2), max_var_q = std::pow(p->max_stddev_q, 2);
        dx = mrgfe_inf_weight(p->var_gain_a, p->fitness_score_thresh, min_var_x, max_var_x, fitness_score);
        dq = mrgfe_inf_weight(p->var_gain_a, p->fitness_score_thresh, min_var_q, max_var_q, fitness_score);
    }
    for (int k = 0; k < 36; ++k) inf[k] = 0.0;
    for (int k = 0; k < 3; ++k) { inf[k * 7] = 1.0 / dx; inf[(k + 3) * 7] = 1.0 / dq; }
    return MRGFE_OK;
}
int mrgfe_calc_information_matrix(mrgfe_ctx* ctx, const mrgfe_inf_params* p, const float* cloud1, size_t n1, const float* cloud2, size_t n2, size_t stride,
                                  const double relpose[16], double inf[36], double* fitness_out)
{
    MRGFE_TRY(check_count(n1, "mrgfe_calc_information_matrix"));
    MRGFE_TRY(check_count(n2, "mrgfe_calc_information_matrix"));
    if (!p || !inf) { set_error("mrgfe_calc_information_matrix: NULL argument"); return MRGFE_ERR_INVALID; }
    double fit = 0.0;
    if (!p->use_const_inf_matrix) MRGFE_TRY(mrgfe_calc_fitness_score(ctx, cloud1, n1, cloud2, n2, stride, relpose, DBL_MAX, &fit));  // the header's default max_range
    if (fitness_out) *fitness_out = fit;
    return mrgfe_inf_matrix_from_fitness(p, fit, inf);
}

// ---- map cloud, other-robot removal, deskewing --------------------------------------------------------------------
// shared tail of the two map-cloud entry points: run the device pass over `total` points (d_cat, or the per-keyframe
// pointers kf_ptrs), apply the reference's emptiness / capacity rules and bring the result down
static int map_cloud_finish(mrgfe_ctx* ctx, int K_all, const float4* d_cat, const float4* const* kf_ptrs, const std::vector<uint32_t>& off, const std::vector<float>& pose_f,
                            float resolution, int min_points_per_voxel, float distance_far_thresh, float* out, size_t capacity, size_t* out_n)
{
    const uint64_t total = off.back();
    size_t m = 0, unfiltered = 0;
    int    rc = MRGFE_OK;
    DevBuf dout;
    if (total) {
        rc = dout.ensure(total * 16);
        if (rc == MRGFE_OK)
            rc = map_cloud_device(ctx, d_cat, off.data(), pose_f.data(), static_cast<int>(off.size()) - 1, resolution, min_points_per_voxel, distance_far_thresh, dout.as<float4>(), &m,
                                  &unfiltered, kf_ptrs);
    }
    // :57-60: the cloud BEFORE the voxel filter decides
    if (rc == MRGFE_OK && unfiltered == 0 && K_all > 1) { set_error("cloud is empty after processing keyframes"); rc = MRGFE_ERR_EMPTY; }
    if (rc == MRGFE_OK && m > capacity) { *out_n = m; set_error("map cloud: output needs %zu points, capacity is %zu", m, capacity); rc = MRGFE_ERR_INVALID; }
    if (rc == MRGFE_OK && m) {
        if (!out) { set_error("map cloud: NULL output"); rc = MRGFE_ERR_INVALID; }
        else if (hipMemcpyAsync(out, dout.p, m * 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            set_error("map cloud: device to host copy failed");
            rc = MRGFE_ERR_HIP;
        }
    }
    if (rc == MRGFE_OK) *out_n = m;
    dout.release();
    return rc;
}

int mrgfe_map_cloud_generate(mrgfe_ctx* ctx, int K, const float* const* clouds, const size_t* n_points, size_t stride, const double* poses, const uint8_t* first_keyframe,
                             float resolution, int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud, float* out, size_t capacity, size_t* out_n)
{
    if (!ctx || !out_n || (K > 0 && (!clouds || !n_points || !poses))) { set_error("mrgfe_map_cloud_generate: NULL argument"); return MRGFE_ERR_INVALID; }
    *out_n = 0;
    if (K <= 0) { set_error("keyframes are empty, cannot generate map cloud"); return MRGFE_ERR_EMPTY; }  // map_cloud_generator.cpp:19-22
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    std::vector<uint32_t> off(1, 0u);
    std::vector<float>    pose_f;
    std::vector<int>      used;
    uint64_t total = 0;
    for (int k = 0; k < K; ++k) {
        if (first_keyframe && first_keyframe[k] && skip_first_cloud) continue;  // :32-34
        if (n_points[k] && !clouds[k]) { set_error("mrgfe_map_cloud_generate: NULL cloud %d", k); return MRGFE_ERR_INVALID; }
        total += n_points[k];
        if (total > 0x7fffffffu) { set_error("mrgfe_map_cloud_generate: more than 2^31 points"); return MRGFE_ERR_INVALID; }
        used.push_back(k);
        off.push_back(static_cast<uint32_t>(total));
        for (int t = 0; t < 16; ++t) pose_f.push_back(static_cast<float>(poses[16 * k + t]));  // pose.matrix().cast<float>()
    }
    int    rc = MRGFE_OK;
    DevBuf dcat;
    if (total) {
        rc = dcat.ensure(total * 16);
        for (size_t u = 0; u < used.size() && rc == MRGFE_OK; ++u)
            if (n_points[used[u]]) rc = upload_cloud(ctx, clouds[used[u]], n_points[used[u]], stride, dcat.as<char>() + size_t(off[u]) * 16);
    }
    if (rc == MRGFE_OK) rc = map_cloud_finish(ctx, K, dcat.as<float4>(), nullptr, off, pose_f, resolution, min_points_per_voxel, distance_far_thresh, out, capacity, out_n);
    dcat.release();
    return rc;
}

// ---- map store: keyframe clouds resident in HBM (include/mrgfe.h) --------------------------------------------------------
struct mrgfe_map_store {
    mrgfe_ctx* ctx = nullptr;
    Arena      arena;  // append-only
    struct Entry { const float4* p; uint32_t n; };
    std::unordered_map<uint64_t, Entry> clouds;
    size_t bytes = 0;
    // exact-NN grids of the keyframes that were `cloud1` of a fitness score lately (graph edges of one keyframe come in bursts:
    // its odometry edge, then the loop edges of the same optimisation cycle), least recently used first out
    struct CachedGrid { uint64_t key = 0; uint64_t tick = 0; NnGrid grid; };
    std::vector<CachedGrid*> grids;
    uint64_t tick = 0;
    size_t   max_grids = 8;
};

int mrgfe_map_store_create(mrgfe_ctx* ctx, mrgfe_map_store** out)
{
    if (!ctx || !out) { set_error("mrgfe_map_store_create: NULL argument"); return MRGFE_ERR_INVALID; }
    mrgfe_map_store* s = new (std::nothrow) mrgfe_map_store();
    if (!s) { set_error("out of host memory"); return MRGFE_ERR_INVALID; }
    s->ctx = ctx;
    *out = s;
    return MRGFE_OK;
}
void mrgfe_map_store_destroy(mrgfe_map_store* s)
{
    if (!s) return;
    {
        MRGFE_LOCK(s->ctx);
        (void)s->ctx->bind();
        for (auto* g : s->grids) { g->grid.release(); delete g; }
        s->arena.release();
    }
    delete s;
}
int mrgfe_map_store_add(mrgfe_map_store* s, uint64_t key, const float* xyzi, size_t n, size_t stride)
{
    MRGFE_TRY(check_count(n, "mrgfe_map_store_add"));
    if (!s || key == 0 || (n && !xyzi)) { set_error("mrgfe_map_store_add: NULL store / cloud or key 0"); return MRGFE_ERR_INVALID; }
    if (n > 0x7fffffffu) { set_error("cloud too large"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(s->ctx);
    MRGFE_TRY(s->ctx->bind());
    auto it = s->clouds.find(key);
    if (it != s->clouds.end()) {
        if (it->second.n == n) return MRGFE_OK;
        set_error("mrgfe_map_store_add: keyframe %llu is stored with %u points, not %zu", static_cast<unsigned long long>(key), it->second.n, n);
        return MRGFE_ERR_INVALID;
    }
    void* p = nullptr;
    if (n) {
        MRGFE_TRY(s->arena.alloc(n * 16, &p));
        MRGFE_TRY(upload_cloud(s->ctx, xyzi, n, stride, p));
    }
    s->clouds[key] = {static_cast<const float4*>(p), static_cast<uint32_t>(n)};
    s->bytes += n * 16;
    return MRGFE_OK;
}
int mrgfe_map_store_has(const mrgfe_map_store* s, uint64_t key, size_t* n)
{
    if (!s) return 0;
    MRGFE_LOCK(s->ctx);
    auto it = s->clouds.find(key);
    if (it == s->clouds.end()) return 0;
    if (n) *n = it->second.n;
    return 1;
}
size_t mrgfe_map_store_bytes(const mrgfe_map_store* s)
{
    if (!s) return 0;
    MRGFE_LOCK(s->ctx);
    return s->bytes;
}
int mrgfe_map_store_generate(mrgfe_map_store* s, int K, const uint64_t* keys, const double* poses, const uint8_t* first_keyframe, float resolution, int min_points_per_voxel,
                             float distance_far_thresh, int skip_first_cloud, float* out, size_t capacity, size_t* out_n)
{
    if (!s || !out_n || (K > 0 && (!keys || !poses))) { set_error("mrgfe_map_store_generate: NULL argument"); return MRGFE_ERR_INVALID; }
    *out_n = 0;
    if (K <= 0) { set_error("keyframes are empty, cannot generate map cloud"); return MRGFE_ERR_EMPTY; }  // map_cloud_generator.cpp:19-22
    MRGFE_LOCK(s->ctx);
    MRGFE_TRY(s->ctx->bind());
    std::vector<uint32_t>      off(1, 0u);
    std::vector<float>         pose_f;
    std::vector<const float4*> ptrs;
    uint64_t total = 0;
    for (int k = 0; k < K; ++k) {
        if (first_keyframe && first_keyframe[k] && skip_first_cloud) continue;  // :32-34
        auto it = s->clouds.find(keys[k]);
        if (it == s->clouds.end()) { set_error("mrgfe_map_store_generate: keyframe %llu is not in the store", static_cast<unsigned long long>(keys[k])); return MRGFE_ERR_INVALID; }
        total += it->second.n;
        if (total > 0x7fffffffu) { set_error("mrgfe_map_store_generate: more than 2^31 points"); return MRGFE_ERR_INVALID; }
        ptrs.push_back(it->second.p);
        off.push_back(static_cast<uint32_t>(total));
        for (int t = 0; t < 16; ++t) pose_f.push_back(static_cast<float>(poses[16 * k + t]));
    }
    MRGFE_HIP_CHECK(hipStreamSynchronize(s->ctx->stream));  // clouds added just before are still on their way up
    return map_cloud_finish(s->ctx, K, nullptr, ptrs.data(), off, pose_f, resolution, min_points_per_voxel, distance_far_thresh, out, capacity, out_n);
}

int mrgfe_map_store_fitness(mrgfe_map_store* s, uint64_t key1, uint64_t key2, const double relpose[16], double max_range, double* out)
{
    if (!s || !relpose || !out) { set_error("mrgfe_map_store_fitness: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(s->ctx);
    MRGFE_TRY(s->ctx->bind());
    auto i1 = s->clouds.find(key1), i2 = s->clouds.find(key2);
    if (i1 == s->clouds.end() || i2 == s->clouds.end()) {
        set_error("mrgfe_map_store_fitness: keyframe %llu is not in the store", static_cast<unsigned long long>(i1 == s->clouds.end() ? key1 : key2));
        return MRGFE_ERR_INVALID;
    }
    if (i1->second.n == 0 || i2->second.n == 0) { *out = DBL_MAX; return MRGFE_OK; }
    mrgfe_map_store::CachedGrid* g = nullptr;
    for (auto* c : s->grids) if (c->key == key1) g = c;
    if (!g) {
        if (s->grids.size() < s->max_grids) { g = new mrgfe_map_store::CachedGrid(); s->grids.push_back(g); }
        else { g = s->grids[0]; for (auto* c : s->grids) if (c->tick < g->tick) g = c; }
        g->key = 0;
        MRGFE_TRY(g->grid.build(s->ctx, i1->second.p, i1->second.n, 1.0f, NnGrid::kCrowding1nn, 1));
        g->key = key1;
    }
    g->tick = ++s->tick;
    float T[16];  // relpose.cast<float>(), row-major
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T[r * 4 + c] = static_cast<float>(relpose[c * 4 + r]);
    return g->grid.fitness(s->ctx, i2->second.p, i2->second.n, T, max_range, out);
}
int mrgfe_map_store_information_matrix(mrgfe_map_store* s, const mrgfe_inf_params* p, uint64_t key1, uint64_t key2, const double relpose[16], double inf[36], double* fitness_out)
{
    if (!p || !inf) { set_error("mrgfe_map_store_information_matrix: NULL argument"); return MRGFE_ERR_INVALID; }
    double fit = 0.0;
    if (!p->use_const_inf_matrix) MRGFE_TRY(mrgfe_map_store_fitness(s, key1, key2, relpose, DBL_MAX, &fit));
    if (fitness_out) *fitness_out = fit;
    return mrgfe_inf_matrix_from_fitness(p, fit, inf);
}

int mrgfe_remove_points_near(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, const float* centres, int n_centres, float radius_sqr, float* kept, size_t* n_kept,
                             float* removed, size_t* n_removed)
{
    MRGFE_TRY(check_count(n, "mrgfe_remove_points_near"));
    if (!ctx || !n_kept || (n && (!xyzi || !kept)) || (n_centres > 0 && !centres) || n_centres < 0) { set_error("mrgfe_remove_points_near: bad argument"); return MRGFE_ERR_INVALID; }
    *n_kept = 0;
    if (n_removed) *n_removed = 0;
    if (n == 0) return MRGFE_OK;
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    DevBuf din, dk, dr;
    int rc = din.ensure(n * 16);
    if (rc == MRGFE_OK) rc = dk.ensure(n * 16);
    if (rc == MRGFE_OK && removed) rc = dr.ensure(n * 16);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, xyzi, n, stride, din.p);
    size_t nk = 0, nr = 0;
    if (rc == MRGFE_OK) rc = remove_points_near_device(ctx, din.as<float4>(), n, centres, n_centres, radius_sqr, dk.as<float4>(), &nk, removed ? dr.as<float4>() : nullptr, &nr);
    if (rc == MRGFE_OK && nk && hipMemcpyAsync(kept, dk.p, nk * 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = MRGFE_ERR_HIP;
    if (rc == MRGFE_OK && removed && nr && hipMemcpyAsync(removed, dr.p, nr * 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = MRGFE_ERR_HIP;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == MRGFE_OK) rc = MRGFE_ERR_HIP;
    if (rc == MRGFE_ERR_HIP) set_error("mrgfe_remove_points_near: device to host copy failed");
    if (rc == MRGFE_OK) { *n_kept = nk; if (n_removed) *n_removed = nr; }
    din.release(); dk.release(); dr.release();
    return rc;
}

int mrgfe_deskew(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, const float ang_v[3], double scan_period, float* out)
{
    MRGFE_TRY(check_count(n, "mrgfe_deskew"));
    if (!ctx || !ang_v || (n && (!xyzi || !out))) { set_error("mrgfe_deskew: NULL argument"); return MRGFE_ERR_INVALID; }
    if (n == 0) return MRGFE_OK;
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    DevBuf din, dout;
    int rc = din.ensure(n * 16);
    if (rc == MRGFE_OK) rc = dout.ensure(n * 16);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, xyzi, n, stride, din.p);
    if (rc == MRGFE_OK) rc = deskew_device(ctx, din.as<float4>(), n, ang_v, scan_period, dout.as<float4>());
    if (rc == MRGFE_OK && (hipMemcpyAsync(out, dout.p, n * 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        set_error("mrgfe_deskew: device to host copy failed");
        rc = MRGFE_ERR_HIP;
    }
    din.release(); dout.release();
    return rc;
}

int mrgfe_transform_cloud(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride, const float T[16], float* out)
{
    MRGFE_TRY(check_count(n, "mrgfe_transform_cloud"));
    if (!ctx || !T || (n && (!xyzi || !out))) { set_error("mrgfe_transform_cloud: NULL argument"); return MRGFE_ERR_INVALID; }
    if (n == 0) return MRGFE_OK;
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    float Tr[16];
    col2row(T, Tr);
    DevBuf din, dout;
    int rc = din.ensure(n * 16);
    if (rc == MRGFE_OK) rc = dout.ensure(n * 16);
    if (rc == MRGFE_OK) rc = upload_cloud(ctx, xyzi, n, stride, din.p);
    if (rc == MRGFE_OK) rc = transform_cloud_device(ctx, din.as<float4>(), n, Tr, dout.as<float4>());
    if (rc == MRGFE_OK && (hipMemcpyAsync(out, dout.p, n * 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        set_error("mrgfe_transform_cloud: device to host copy failed");
        rc = MRGFE_ERR_HIP;
    }
    din.release(); dout.release();
    return rc;
}

// ---- batch ------------------------------------------------------------------------------------------------------
int mrgfe_batch_create(mrgfe_ctx* ctx, const mrgfe_reg_params* params, mrgfe_batch** out)
{
    if (!ctx || !out) { set_error("mrgfe_batch_create: NULL argument"); return MRGFE_ERR_INVALID; }
    *out = nullptr;
    MRGFE_TRY(check_params(params));
    if (params->method == MRGFE_ICP_HIP || params->method == MRGFE_PCL_GICP_HIP || params->method == MRGFE_PCL_GICP_OMP_HIP) {
        set_error("mrgfe_batch_create: ICP_HIP and PCL_GICP_HIP are offered for single registrations only");
        return MRGFE_ERR_INVALID;
    }
    mrgfe_batch* b = new (std::nothrow) mrgfe_batch();
    if (!b) { set_error("out of host memory"); return MRGFE_ERR_INVALID; }
    b->ctx = ctx;
    b->params = *params;
    b->ndt = new NdtEngine(ctx, ndt_params_from(*params));
    if (const char* e = std::getenv("MRGFE_KEYFRAME_STORE_MB")) b->store_cap = static_cast<size_t>(std::max(0.0, std::atof(e))) << 20;
    *out = b;
    return MRGFE_OK;
}
static void batch_async_main(mrgfe_batch* b)
{
    mrgfe_batch::Async& a = *b->async;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(a.mu);
            a.cv.wait(lk, [&] { return a.state == 1 || a.quit; });
            if (a.quit) return;
        }
        int st;
        std::string err;
        {
            MRGFE_LOCK(b->ctx);
            {
                std::lock_guard<std::mutex> lk(a.mu);
                a.state = 2;
            }
            a.cv.notify_all();
            try {
                st = mrgfe_batch_align(b, a.fitness_max_range, a.results);
                if (st != MRGFE_OK) err = mrgfe_last_error();
            } catch (const std::exception& e) {  // (a host container's bad_alloc: an error code for the waiter, never std::terminate)
                st = MRGFE_ERR_INVALID;
                err = std::string("mrgfe_batch_align_async: ") + e.what();
            } catch (...) {
                st = MRGFE_ERR_INVALID;
                err = "mrgfe_batch_align_async: unknown exception";
            }
        }
        {
            std::lock_guard<std::mutex> lk(a.mu);
            a.status = st;
            a.error = err;
            a.state = 3;
        }
        a.cv.notify_all();
    }
}

int mrgfe_batch_align_async(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results)
{
    if (!b || !results) { set_error("mrgfe_batch_align_async: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!b->async) {
        b->async.reset(new (std::nothrow) mrgfe_batch::Async());
        if (!b->async) { set_error("out of host memory"); return MRGFE_ERR_INVALID; }
        try {
            b->async->th = std::thread(batch_async_main, b);
        } catch (const std::exception& e) {
            b->async.reset();
            set_error("mrgfe_batch_align_async: %s", e.what());
            return MRGFE_ERR_INVALID;
        }
    }
    mrgfe_batch::Async& a = *b->async;
    std::unique_lock<std::mutex> lk(a.mu);
    if (a.state != 0) { set_error("mrgfe_batch_align_async: an align of this batch is %s: call mrgfe_batch_wait first", a.state == 3 ? "finished and not yet waited for" : "in flight"); return MRGFE_ERR_STATE; }
    a.fitness_max_range = fitness_max_range;
    a.results = results;
    a.state = 1;
    a.cv.notify_all();
    a.cv.wait(lk, [&] { return a.state >= 2; });  // the worker holds the context lock now: later calls on this batch queue up behind the align
    return MRGFE_OK;
}

int mrgfe_batch_wait(mrgfe_batch* b)
{
    if (!b) { set_error("mrgfe_batch_wait: NULL batch"); return MRGFE_ERR_INVALID; }
    if (!b->async) { set_error("mrgfe_batch_wait: no asynchronous align was started"); return MRGFE_ERR_STATE; }
    mrgfe_batch::Async& a = *b->async;
    std::unique_lock<std::mutex> lk(a.mu);
    if (a.state == 0) { set_error("mrgfe_batch_wait: no asynchronous align was started"); return MRGFE_ERR_STATE; }
    a.cv.wait(lk, [&] { return a.state == 3; });
    a.state = 0;
    if (a.status != MRGFE_OK) set_error("%s", a.error.c_str());
    return a.status;
}

void mrgfe_batch_destroy(mrgfe_batch* b)
{
    if (!b) return;
    if (b->async) {  // before the context lock below: a running align holds it
        mrgfe_batch::Async& a = *b->async;
        {
            std::unique_lock<std::mutex> lk(a.mu);
            a.cv.wait(lk, [&] { return a.state == 0 || a.state == 3; });
            a.quit = true;
        }
        a.cv.notify_all();
        if (a.th.joinable()) a.th.join();
    }
    {
        MRGFE_LOCK(b->ctx);
        (void)b->ctx->bind();
        for (auto& g : b->fit_grids) g.release();
        for (auto& gs : b->fit_sets) gs->release();
        for (mrgfe_ctx* fc : b->fit_ctxs) mrgfe_ctx_destroy(fc);
        if (b->early_ctx) mrgfe_ctx_destroy(b->early_ctx);
        if (b->uploads_done) (void)hipEventDestroy(b->uploads_done);
        if (b->port) b->port->buf.release();
        for (auto& gp : b->gicp_pairs) { gp.cov.release(); gp.corr.release(); gp.mahal.release(); }
        delete b->gicp_batch;
        for (auto* g : b->gicp) delete g;
        delete b->ndt;
        for (auto& kv : b->store) { kv.second->cloud.release(); kv.second->cov.release(); delete kv.second; }
    }
    delete b;
}
int mrgfe_batch_clear(mrgfe_batch* b)
{
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    b->ndt->clear();
    for (auto* g : b->gicp) delete g;  // their cached target state belongs to the clouds just forgotten
    b->gicp.clear();
    b->pair_key.clear();
    ++b->epoch;  // stored keyframes stay; none is referenced by the (now empty) batch
    b->t_queue = std::chrono::steady_clock::now();  // mrgfe_batch_timing: the next align's time starts where its queueing starts
    b->t_queue_set = true;
    return MRGFE_OK;
}
int mrgfe_batch_add_target(mrgfe_batch* b, const float* xyzi, size_t n, size_t stride)
{
    MRGFE_TRY(check_count(n, "mrgfe_batch_add_target"));
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    return b->ndt->add_target_host(xyzi, n, stride);
}
int mrgfe_batch_add_target_device(mrgfe_batch* b, const void* d, size_t n)
{
    MRGFE_TRY(check_count(n, "mrgfe_batch_add_target_device"));
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    return b->ndt->add_target_device(d, n);
}
int mrgfe_batch_add_pair(mrgfe_batch* b, int target, const float* xyzi, size_t n, size_t stride, const float guess[16])
{
    MRGFE_TRY(check_count(n, "mrgfe_batch_add_pair"));
    if (!b || !guess) { set_error("NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    float g[16];
    col2row(guess, g);
    return b->ndt->add_pair_host(target, xyzi, n, stride, g);
}
int mrgfe_batch_add_pair_device(mrgfe_batch* b, int target, const void* d, size_t n, const float guess[16])
{
    MRGFE_TRY(check_count(n, "mrgfe_batch_add_pair_device"));
    if (!b || !guess) { set_error("NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    float g[16];
    col2row(guess, g);
    return b->ndt->add_pair_device(target, d, n, g);
}
int mrgfe_batch_add_device(mrgfe_batch* b, int n_targets, const void* const* d_targets, const size_t* target_points, int n_pairs, const int32_t* pair_target,
                           const void* const* d_sources, const size_t* source_points, const float* guesses)
{
    if (!b || n_targets < 0 || n_pairs < 0 || (n_targets && (!d_targets || !target_points)) || (n_pairs && (!pair_target || !d_sources || !source_points || !guesses))) {
        set_error("mrgfe_batch_add_device: bad argument");
        return MRGFE_ERR_INVALID;
    }
    MRGFE_LOCK(b->ctx);
    const int t0 = b->ndt->n_targets(), p0 = b->ndt->n_pairs();
    for (int i = 0; i < n_pairs; ++i)
        if (pair_target[i] < 0 || pair_target[i] >= n_targets) { set_error("mrgfe_batch_add_device: pair %d names target %d of %d", i, pair_target[i], n_targets); return MRGFE_ERR_INVALID; }
    for (int i = 0; i < n_targets; ++i) {
        const int t = b->ndt->add_target_device(d_targets[i], target_points[i]);
        if (t < 0) return t;
    }
    for (int i = 0; i < n_pairs; ++i) {
        float g[16];
        col2row(guesses + size_t(i) * 16, g);
        const int pi = b->ndt->add_pair_device(t0 + pair_target[i], d_sources[i], source_points[i], g);
        if (pi < 0) return pi;
    }
    return p0;
}
int mrgfe_batch_add_pair_keyed(mrgfe_batch* b, int target, uint64_t key, const float* xyzi, size_t n, size_t stride, const float guess[16])
{
    MRGFE_TRY(check_count(n, "mrgfe_batch_add_pair_keyed"));
    if (!b || !guess) { set_error("NULL argument"); return MRGFE_ERR_INVALID; }
    if (key == 0) return mrgfe_batch_add_pair(b, target, xyzi, n, stride, guess);
    MRGFE_LOCK(b->ctx);
    MRGFE_TRY(b->ctx->bind());
    if (n > 0x7fffffffu) { set_error("cloud too large"); return MRGFE_ERR_INVALID; }
    mrgfe_batch::Keyframe* kf = nullptr;
    auto it = b->store.find(key);
    if (it != b->store.end() && it->second->n == n) {
        kf = it->second;
    } else {
        if (n && !xyzi) { set_error("mrgfe_batch_add_pair_keyed: key %llu is not in the store (or has another size) and no cloud was given", static_cast<unsigned long long>(key)); return MRGFE_ERR_INVALID; }
        if (it != b->store.end()) {  // same key, different cloud: replace — unless this batch already uses the old one
            if (it->second->last_epoch == b->epoch) { set_error("mrgfe_batch_add_pair_keyed: key %llu is already used in this batch with %u points", static_cast<unsigned long long>(key), it->second->n); return MRGFE_ERR_INVALID; }
            it->second->cloud.release();
            it->second->cov.release();
            delete it->second;
            b->store.erase(it);
        }
        store_make_room(b, n * 16 + (!is_ndt(b->params.method) ? n * 48 : 0));
        kf = new (std::nothrow) mrgfe_batch::Keyframe();
        if (!kf) { set_error("out of host memory"); return MRGFE_ERR_INVALID; }
        int rc = kf->cloud.ensure(std::max<size_t>(n, 1) * 16);
        if (rc == MRGFE_OK && n) rc = upload_cloud(b->ctx, xyzi, n, stride, kf->cloud.p);
        if (rc != MRGFE_OK) { kf->cloud.release(); delete kf; return rc; }
        kf->n = static_cast<uint32_t>(n);
        b->store[key] = kf;
    }
    kf->last_epoch = b->epoch;
    kf->last_tick = ++b->tick;
    float g[16];
    col2row(guess, g);
    const int pair = b->ndt->add_pair_device(target, kf->cloud.p, n, g);
    if (pair >= 0) {
        if (b->pair_key.size() <= static_cast<size_t>(pair)) b->pair_key.resize(pair + 1, 0);
        b->pair_key[pair] = key;
    }
    return pair;
}
int mrgfe_batch_has_cloud(const mrgfe_batch* b, uint64_t key, size_t* n)
{
    if (!b || key == 0) return 0;
    MRGFE_LOCK(b->ctx);
    auto it = b->store.find(key);
    if (it == b->store.end()) return 0;
    if (n) *n = it->second->n;
    return 1;
}
size_t mrgfe_batch_store_bytes(const mrgfe_batch* b)
{
    if (!b) return 0;
    MRGFE_LOCK(b->ctx);
    size_t total = 0;
    for (auto& kv : b->store) total += kv.second->bytes();
    return total;
}
int mrgfe_batch_forget(mrgfe_batch* b, uint64_t key)
{
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    MRGFE_TRY(b->ctx->bind());
    for (auto it = b->store.begin(); it != b->store.end();) {
        if (key != 0 && it->first != key) { ++it; continue; }
        if (it->second->last_epoch == b->epoch && !b->pair_key.empty()) { set_error("mrgfe_batch_forget: key %llu is used by the current batch (clear it first)", static_cast<unsigned long long>(it->first)); return MRGFE_ERR_STATE; }
        it->second->cloud.release();
        it->second->cov.release();
        delete it->second;
        it = b->store.erase(it);
    }
    return MRGFE_OK;
}
int mrgfe_batch_set_guess(mrgfe_batch* b, int pair, const float guess[16])
{
    if (!b || !guess) { set_error("NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    float g[16];
    col2row(guess, g);
    return b->ndt->set_guess(pair, g);
}
int mrgfe_batch_build_targets(mrgfe_batch* b)
{
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    if (!is_ndt(b->params.method)) return MRGFE_OK;  // GICP variants: target covariances and grids are built by the first align
    TraceRange tr("mrgfe set_target (batch)");
    return b->ndt->build_targets();
}
int mrgfe_batch_num_pairs(const mrgfe_batch* b) { return b ? b->ndt->n_pairs() : 0; }

static int batch_align_impl(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results);

int mrgfe_batch_align(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results)
{
    if (!b || !results) { set_error("mrgfe_batch_align: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    const auto t_start = b->t_queue_set ? b->t_queue : std::chrono::steady_clock::now();
    int st;
    {
        TraceRange tr("mrgfe_batch_align");
        st = batch_align_impl(b, fitness_max_range, results);
    }
    if (st == MRGFE_OK) {
        b->last_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count();
        b->last_pairs = b->ndt->n_pairs();
        b->total_us += b->last_us;
        b->total_pairs += b->last_pairs;
    }
    b->t_queue_set = false;
    // zero-copy uploads (mrgfe_ctx_set_zero_copy_uploads): the caller's page-locked clouds are its own again when this call returns — also when it fails
    // with their DMA still queued.  The error text of the failure is kept.
    if (st != MRGFE_OK && b->ctx->dma_from_caller) {
        const std::string msg = mrgfe_last_error();
        drain_caller_dma(b->ctx);
        set_error("%s", msg.c_str());
    } else {
        b->ctx->dma_from_caller = false;  // a successful align has waited for its stream
    }
    return st;
}

static int batch_align_impl(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results)
{
    MRGFE_LOCK(b->ctx);
    NdtEngine& e = *b->ndt;
    const int P = e.n_pairs();
    const bool gicp = !is_ndt(b->params.method);
    std::vector<char> fit_built;   // targets whose fitness grid is built in this call
    std::vector<char> early_skip;  // pairs whose fitness score was computed beside the alignment rounds
    if (gicp) {
        // GICP_HIP: the candidates of a target share its covariances and correspondence grid, and all LM loops advance
        // together (GicpBatch: one launch per kernel and round for the pairs still running)
        MRGFE_TRY(b->ctx->bind());
        if (b->gicp.size() < static_cast<size_t>(e.n_targets())) b->gicp.resize(e.n_targets(), nullptr);
        if (!b->gicp_batch) b->gicp_batch = new GicpBatch(b->ctx);
        for (size_t i = P; i < b->gicp_pairs.size(); ++i) { b->gicp_pairs[i].cov.release(); b->gicp_pairs[i].corr.release(); b->gicp_pairs[i].mahal.release(); }
        b->gicp_pairs.resize(P);
        for (int i = 0; i < P; ++i) {
            const NdtPairInfo& p = e.pair(i);
            const NdtTargetInfo& t = e.target(p.target);
            GicpEngine*& g = b->gicp[p.target];
            if (!g) {
                g = new GicpEngine(b->ctx, gicp_params_from(b->params));
                MRGFE_TRY(g->set_target(t.d_pts, t.n));
            }
            GicpBatchPair& bp = b->gicp_pairs[i];
            bp.target = p.target;
            bp.d_src = p.d_src;
            bp.n = p.n;
            bp.ext_cov = nullptr;
            bp.ext_cov_k = nullptr;
            if (static_cast<size_t>(i) < b->pair_key.size() && b->pair_key[i]) {
                mrgfe_batch::Keyframe* kf = b->store.at(b->pair_key[i]);
                bp.ext_cov = &kf->cov;
                bp.ext_cov_k = &kf->cov_k;
            }
            std::memcpy(bp.guess, p.guess, sizeof(bp.guess));
        }
        MRGFE_TRY(b->gicp_batch->align_all(b->gicp, b->gicp_pairs));
        b->gicp_final.assign(size_t(P) * 16, 0.0f);
        for (int i = 0; i < P; ++i) {
            const GicpLmController& c = b->gicp_pairs[i].ctl;
            mrgfe_pair_result& r = results[i];
            c.final_transformation(&b->gicp_final[size_t(i) * 16]);
            row2col(&b->gicp_final[size_t(i) * 16], r.T);
            std::memcpy(r.H, c.hessian(), sizeof(r.H));
            r.fitness = DBL_MAX;
            r.trans_probability = 0.0;
            r.converged = c.converged() ? 1 : 0;
            r.iterations = c.iterations();
            r.evaluations = c.evaluations();
            r.pair_id = i;
        }
    } else {
        // getFitnessScore needs an exact-NN grid per distinct target, and those depend on the target clouds only: they are built
        // on a helper context by a second host thread WHILE the alignment rounds run (their small launches fill the tails of the
        // derivative kernels), instead of one after the other behind the alignment (64 targets: ~10 ms of a 65 ms step)
        // One grid is a dozen small launches and half a dozen host waits (bounding box, the adaptive cell size, the scan table): 64
        // of them in a row took 30 ms of wall time for 4 ms of kernels and outlasted the 17 ms of alignment they were meant to
        // hide behind.  The targets are dealt to up to four builder threads, each with its own context.
        std::vector<std::thread> builders;
        int          build_status = MRGFE_OK;
        std::string  build_error;
        std::mutex   build_mu;
        std::vector<int> todo;  // (outlives the threads: they are joined below)
        const bool   overlap = fitness_max_range >= 0 && P >= 2 && std::getenv("MRGFE_NO_FIT_OVERLAP") == nullptr;
        std::unique_ptr<std::atomic<char>[]> grid_ready;  // per target: its fitness grid is complete (set by the builder that made it)
        auto fail = [&](int st, const std::string& why) { std::lock_guard<std::mutex> g(build_mu); if (build_status == MRGFE_OK) { build_status = st; build_error = why; } };
        // Everything that can fail with an early return happens BEFORE the first helper thread exists: a joinable std::thread
        // destroyed by a return would end the process (the SLAM node) instead of reporting the error.
        // Early fitness pass.  Round 4 ran a wave whenever a sixth of the pairs had finished: the chip is still full of derivative work then, and
        // the waves only added their fixed costs (config[3], 256 pairs: 27.8 ms without them, 28.7 ms with).  What IS idle is the tail: a few
        // stragglers line-searching through tens of small rounds (one pair: ~12 us of derivative work on a chip that holds twenty times that).
        // So ONE pass, started when the pairs still running drop to an eighth of the batch (MRGFE_EARLY_FIT_ACTIVE_DIV), scores every finished
        // pair on a helper context beside the stragglers' rounds; the stragglers are scored behind the last round as before.  The count comes
        // from the round plans the device already publishes (no extra kernel until the one snapshot that carries the final transformations).
        int early_min_pairs = 8;
        if (const char* env = std::getenv("MRGFE_EARLY_FIT_MIN_PAIRS")) early_min_pairs = std::max(2, std::atoi(env));
        int early_div = 8;
        if (const char* env = std::getenv("MRGFE_EARLY_FIT_ACTIVE_DIV")) early_div = std::max(1, std::atoi(env));
        const bool early_on = overlap && P >= early_min_pairs && std::getenv("MRGFE_NO_EARLY_FIT") == nullptr;
        if (!b->port) b->port.reset(new NdtSnapshotPort());
        NdtSnapshotPort& port = *b->port;  // (its pinned buffer is kept between calls)
        if (early_on) MRGFE_TRY(port.buf.ensure(sizeof(NdtSnapshotHead) + sizeof(NdtSnapshotRec) * size_t(P)));
        if (overlap) {
            if (fit_built.size() < static_cast<size_t>(e.n_targets())) fit_built.resize(e.n_targets(), 0);
            if (b->fit_grids.size() < static_cast<size_t>(e.n_targets())) b->fit_grids.resize(e.n_targets());
            grid_ready.reset(new std::atomic<char>[std::max(1, e.n_targets())]);
            for (int t = 0; t < e.n_targets(); ++t) grid_ready[t].store(0, std::memory_order_relaxed);
            for (int i = 0; i < P; ++i) {
                const NdtPairInfo& p = e.pair(i);
                if (e.target(p.target).n == 0 || p.n == 0 || fit_built[p.target]) continue;
                fit_built[p.target] = 1;
                todo.push_back(p.target);
            }
            // ... and each thread builds its targets a chunk at a time, every step of the build one launch over the chunk (NnGridSet)
            // ONE builder for up to eight chunks, two beyond: a chunk's build is a dozen launches over all its targets, and a second thread's launches only
            // compete with the first's and with the rounds for the queues (config[3], 64 targets = 4 chunks: 25.9 - 26.2 ms per step with two builders,
            // 25.0 - 25.2 with one; the chunk size makes no difference from 8 to 64 targets)
            size_t n_builders = 0, chunk = 16;
            if (const char* env = std::getenv("MRGFE_FIT_BUILDERS")) n_builders = static_cast<size_t>(std::max(1, std::atoi(env)));
            if (const char* env = std::getenv("MRGFE_FIT_CHUNK")) chunk = static_cast<size_t>(std::max(1, std::atoi(env)));
            const size_t n_chunks = (todo.size() + chunk - 1) / chunk;
            if (n_builders == 0) n_builders = n_chunks > 8 ? 2 : 1;
            n_builders = std::min(n_builders, n_chunks);
            while (b->fit_sets.size() < n_chunks) b->fit_sets.emplace_back(new NnGridSet());
            while (b->fit_ctxs.size() < n_builders) {
                mrgfe_ctx* fc = nullptr;
                if (ctx_create_like(b->ctx, &fc) != MRGFE_OK) return MRGFE_ERR_HIP;  // (same compute-unit mask as the batch's own context)
                b->fit_ctxs.push_back(fc);
            }
            // the early fitness pass runs on a context of its own whose streams have the device's LOWEST priority: the stragglers' small launches
            // on the batch's stream are dispatched ahead of the pass's workgroups as slots free up (at equal priority the tail's rounds took twice as
            // long beside the pass: what the overlap gained, the rounds lost)
            if (early_on && !b->early_ctx && ctx_create_like(b->ctx, &b->early_ctx, -1) != MRGFE_OK) return MRGFE_ERR_HIP;
            // the target clouds reach the device by asynchronous copies (and gathers) on the batch's stream: the helper streams
            // must not read them before those have finished
            MRGFE_TRY(b->ctx->bind());
            if (!b->uploads_done) MRGFE_HIP_CHECK(hipEventCreateWithFlags(&b->uploads_done, hipEventDisableTiming));
            MRGFE_HIP_CHECK(hipEventRecord(b->uploads_done, b->ctx->stream));
            for (size_t w = 0; w < n_builders; ++w)
                builders.emplace_back([b, &e, &todo, w, n_builders, n_chunks, chunk, &fail, &grid_ready] {
                    mrgfe_ctx* fc = b->fit_ctxs[w];
                    std::lock_guard<std::recursive_mutex> lock(fc->mu);
                    if (fc->bind() != MRGFE_OK) { fail(MRGFE_ERR_HIP, mrgfe_last_error()); return; }
                    if (hipStreamWaitEvent(fc->stream, b->uploads_done, 0) != hipSuccess) { fail(MRGFE_ERR_HIP, "helper stream could not wait for the uploads"); return; }
                    std::vector<const float4*> clouds;
                    std::vector<uint32_t>      sizes;
                    std::vector<NnGrid*>       views;
                    for (size_t c = w; c < n_chunks; c += n_builders) {
                        const size_t k0 = c * chunk, k1 = std::min(todo.size(), k0 + chunk);
                        clouds.clear(); sizes.clear(); views.clear();
                        for (size_t k = k0; k < k1; ++k) {
                            const NdtTargetInfo& T = e.target(todo[k]);
                            clouds.push_back(T.d_pts);
                            sizes.push_back(static_cast<uint32_t>(T.n));
                            views.push_back(&b->fit_grids[todo[k]]);
                        }
                        const int st = b->fit_sets[c]->build(fc, clouds.data(), sizes.data(), static_cast<int>(k1 - k0), 1.0f, NnGrid::kCrowding1nn, 1, views.data());  // returns with the grids complete (synchronised)
                        if (st != MRGFE_OK) { fail(st, mrgfe_last_error()); return; }
                        for (size_t k = k0; k < k1; ++k) grid_ready[todo[k]].store(1, std::memory_order_release);
                    }
                    if (hipStreamSynchronize(fc->stream) != hipSuccess) fail(MRGFE_ERR_HIP, "helper stream synchronisation failed");
                });
        }
        // Early fitness passes.  The rounds of a batch end in a long tail — a few stragglers line-searching while most alignments have
        // finished and the chip idles between their small launches — and getFitnessScore of a finished pair needs nothing but its final
        // transformation.  A second host thread asks the aligning thread for snapshots (NdtSnapshotPort), and whenever enough finished
        // pairs with a complete grid have accumulated it runs their passes on a helper context beside the remaining rounds.  A pair's
        // score does not depend on the launch it is computed in (nn_fit_sum_kernel's fixed slices), so the records are the same.
        port.want.store(0);
        port.issued.store(0);
        port.finished.store(0);
        std::vector<char> early_done(P, 0);
        std::thread early;
        b->fit_total = FitStats();
        if (early_on) {
            port.head()->tag = 0;
            port.n_active.store(static_cast<uint32_t>(P), std::memory_order_release);
            early = std::thread([&, P, early_div] {
                mrgfe_ctx* fc = b->early_ctx;
                std::lock_guard<std::recursive_mutex> lock(fc->mu);
                if (fc->bind() != MRGFE_OK) { fail(MRGFE_ERR_HIP, mrgfe_last_error()); return; }
                const uint32_t threshold = static_cast<uint32_t>(std::max(1, P / early_div));
                // wait for the tail (or the end of the alignment)
                while (!port.finished.load(std::memory_order_acquire) && port.n_active.load(std::memory_order_acquire) > threshold) std::this_thread::sleep_for(std::chrono::microseconds(20));
                if (port.finished.load(std::memory_order_acquire)) return;
                port.want.store(1, std::memory_order_release);
                while (port.issued.load(std::memory_order_acquire) == 0 && !port.finished.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(5));
                const uint32_t tag = port.issued.load(std::memory_order_acquire);
                if (tag == 0) return;  // finished without a snapshot
                volatile NdtSnapshotHead* hd = port.head();
                while (__atomic_load_n(&hd->tag, __ATOMIC_ACQUIRE) != tag) {
                    if (port.finished.load(std::memory_order_acquire) && __atomic_load_n(&hd->tag, __ATOMIC_ACQUIRE) != tag) return;  // align_all failed before the kernel ran
                    std::this_thread::sleep_for(std::chrono::microseconds(5));
                }
                std::vector<NnFitnessJob> jobs;
                std::vector<int>          job_pair;
                const NdtSnapshotRec* recs = port.recs();
                for (int i = 0; i < P; ++i) {
                    if (!recs[i].done) continue;
                    const NdtPairInfo& p = e.pair(i);
                    if (e.target(p.target).n == 0 || p.n == 0 || !grid_ready[p.target].load(std::memory_order_acquire)) continue;
                    float T[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
                    std::memcpy(T, recs[i].T12, sizeof(recs[i].T12));
                    jobs.push_back(b->fit_grids[p.target].make_fitness_job(p.d_src, p.n, T));
                    job_pair.push_back(i);
                }
                if (jobs.empty()) return;
                std::vector<double> fit(jobs.size(), 0.0);
                TraceRange tr("mrgfe early fitness pass");
                const int st = nn_fitness_batch(fc, jobs.data(), jobs.size(), fitness_max_range, fit.data());
                if (st != MRGFE_OK) { fail(st, mrgfe_last_error()); return; }
                b->fit_total.add(fc->fit_stats);
                for (size_t j = 0; j < jobs.size(); ++j) { results[job_pair[j]].fitness = fit[j]; early_done[job_pair[j]] = 1; }
            });
        }
        int align_status;
        {
            TraceRange tr("mrgfe rounds (set_target + align_all)");
            align_status = e.align_all(early_on ? &port : nullptr);
        }
        if (early.joinable()) early.join();
        for (std::thread& t : builders) t.join();
        MRGFE_TRY(align_status);
        if (build_status != MRGFE_OK) { set_error("%s", build_error.c_str()); return build_status; }
        for (int i = 0; i < P; ++i) {
            const NdtController& c = e.pair(i).ctl;
            mrgfe_pair_result& r = results[i];
            row2col(c.final_transformation(), r.T);
            std::memcpy(r.H, c.hessian(), sizeof(r.H));
            if (!early_done[i]) r.fitness = DBL_MAX;
            r.trans_probability = c.trans_probability();
            r.converged = c.converged() ? 1 : 0;
            r.iterations = c.iterations();
            r.evaluations = c.evaluations();
            r.pair_id = i;
        }
        early_skip.swap(early_done);
    }
    if (fitness_max_range >= 0) {
        TraceRange tr("mrgfe fitness passes");
        // getFitnessScore of every pair in one launch: one exact-NN grid per distinct target
        std::vector<NnGrid>& grids = b->fit_grids;
        if (grids.size() < static_cast<size_t>(e.n_targets())) grids.resize(e.n_targets());
        if (fit_built.size() < static_cast<size_t>(e.n_targets())) fit_built.resize(e.n_targets(), 0);
        std::vector<char>&  built = fit_built;
        std::vector<NnFitnessJob> jobs;
        std::vector<int>          job_pair;
        int st = MRGFE_OK;
        for (int i = 0; i < P && st == MRGFE_OK; ++i) {
            const NdtPairInfo& p = e.pair(i);
            const NdtTargetInfo& t = e.target(p.target);
            if (t.n == 0 || p.n == 0 || (static_cast<size_t>(i) < early_skip.size() && early_skip[i])) continue;
            if (!built[p.target]) { st = grids[p.target].build(b->ctx, t.d_pts, t.n, 1.0f, NnGrid::kCrowding1nn, 1); built[p.target] = 1; }
            if (st == MRGFE_OK) { jobs.push_back(grids[p.target].make_fitness_job(p.d_src, p.n, gicp ? &b->gicp_final[size_t(i) * 16] : p.ctl.final_transformation())); job_pair.push_back(i); }
        }
        if (st == MRGFE_OK && !jobs.empty()) {
            std::vector<double> fit(jobs.size());
            st = nn_fitness_batch(b->ctx, jobs.data(), jobs.size(), fitness_max_range, fit.data());
            if (st == MRGFE_OK) {
                b->fit_total.add(b->ctx->fit_stats);
                for (size_t j = 0; j < jobs.size(); ++j) results[job_pair[j]].fitness = fit[j];
            }
        }
        MRGFE_TRY(st);
    }
    return MRGFE_OK;
}

int mrgfe_batch_fitness_stats(const mrgfe_batch* b, double out[11])
{
    if (!b || !out) { set_error("mrgfe_batch_fitness_stats: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    const FitStats& f = b->fit_total;
    const double v[11] = {f.ms_block, f.ms_sweep, f.ms_far, double(f.queries), double(f.queued), double(f.queued_far), double(f.words), double(f.tested), double(f.cells), double(f.points), double(f.calls)};
    std::memcpy(out, v, sizeof(v));
    return MRGFE_OK;
}

int mrgfe_batch_timing(const mrgfe_batch* b, double out[4])
{
    if (!b || !out) { set_error("mrgfe_batch_timing: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    out[0] = b->total_pairs > 0 ? b->total_us / double(b->total_pairs) : 0.0;  // average_time_per_candidate_us (apps/mrg_slam_component.cpp:1032-1037)
    out[1] = b->last_us;
    out[2] = double(b->last_pairs);
    out[3] = double(b->total_pairs);
    return MRGFE_OK;
}
int mrgfe_batch_timing_reset(mrgfe_batch* b)
{
    if (!b) { set_error("mrgfe_batch_timing_reset: NULL batch"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(b->ctx);
    b->last_us = b->total_us = 0.0;
    b->last_pairs = b->total_pairs = 0;
    return MRGFE_OK;
}

int mrgfe_batch_kernel_stats(const mrgfe_batch* b, int mode, double* ms, int64_t* launches, double* bytes)
{
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    b->ndt->kernel_stats(mode, ms, launches, bytes);
    return MRGFE_OK;
}

int mrgfe_dbg_set_prefilter_device_driven(int mode) { return prefilter_set_device_driven(mode); }
int mrgfe_dbg_set_pclgicp_reference_order(int mode) { return gicp_set_pcl_reference_order(mode); }

int mrgfe_batch_largest_launch(const mrgfe_batch* b, double out[4])
{
    if (!b || !out) { set_error("mrgfe_batch_largest_launch: NULL argument"); return MRGFE_ERR_INVALID; }
    out[0] = b->ndt->largest_ms;
    for (int m = 0; m < 3; ++m) out[1 + m] = double(b->ndt->largest_pairs[m]);
    return MRGFE_OK;
}

int mrgfe_batch_pair_counts(const mrgfe_batch* b, int mode, double* points, double* neighbours)
{
    if (!b) { set_error("NULL batch"); return MRGFE_ERR_INVALID; }
    double p = 0, n = 0;
    for (int m = 0; m < 3; ++m)
        if (mode < 0 || mode == m) { p += b->ndt->mode_points[m]; n += b->ndt->mode_neighbours[m]; }
    if (points) *points = p;
    if (neighbours) *neighbours = n;
    return MRGFE_OK;
}

// ---- diagnostics ----------------------------------------------------------------------------------------------
int mrgfe_dbg_sort_pairs(mrgfe_ctx* ctx, const uint32_t* keys, const uint32_t* vals, size_t n, int key_bits, uint32_t* out_keys, uint32_t* out_vals)
{
    MRGFE_TRY(check_count(n, "mrgfe_dbg_sort_pairs"));
    if (!ctx || (n && (!keys || !vals || !out_keys || !out_vals))) { set_error("mrgfe_dbg_sort_pairs: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    if (n == 0) return MRGFE_OK;
    uint32_t  nn = static_cast<uint32_t>(n);
    SliceTable tab;
    tab.build(&nn, 1);
    DevBuf &dk = ctx->scratch[2], &dv = ctx->scratch[3], &dkt = ctx->scratch[4], &dvt = ctx->scratch[5], &dh = ctx->scratch[6], &ds = ctx->scratch[0];
    MRGFE_TRY(dk.ensure(n * 4)); MRGFE_TRY(dv.ensure(n * 4)); MRGFE_TRY(dkt.ensure(n * 4)); MRGFE_TRY(dvt.ensure(n * 4));
    MRGFE_TRY(dh.ensure(sizeof(uint32_t) * 256 * (tab.total_blks + 1)));
    MRGFE_TRY(ds.ensure(sizeof(Slice)));
    MRGFE_HIP_CHECK(hipMemcpy(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice));
    MRGFE_HIP_CHECK(hipMemcpy(dk.p, keys, n * 4, hipMemcpyHostToDevice));
    MRGFE_HIP_CHECK(hipMemcpy(dv.p, vals, n * 4, hipMemcpyHostToDevice));
    uint32_t *sk, *sv;
    MRGFE_TRY(radix_sort_pairs(ctx, dk.as<uint32_t>(), dv.as<uint32_t>(), dkt.as<uint32_t>(), dvt.as<uint32_t>(), ds.as<Slice>(), tab, key_bits, dh.as<uint32_t>(), &sk, &sv));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    MRGFE_HIP_CHECK(hipMemcpy(out_keys, sk, n * 4, hipMemcpyDeviceToHost));
    MRGFE_HIP_CHECK(hipMemcpy(out_vals, sv, n * 4, hipMemcpyDeviceToHost));
    return MRGFE_OK;
}

int mrgfe_dbg_wave_sums(mrgfe_ctx* ctx, int n_vals, const double* in, int cases, double* out_fold, double* out_plain)
{
    if (!ctx || !in || !out_fold || !out_plain || cases < 0) { set_error("mrgfe_dbg_wave_sums: bad argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    const size_t n_in = size_t(cases) * 64 * n_vals, n_out = size_t(cases) * n_vals;
    DevBuf &di = ctx->scratch[0], &df = ctx->scratch[1], &dp = ctx->scratch[2];
    MRGFE_TRY(di.ensure(std::max<size_t>(n_in, 1) * 8)); MRGFE_TRY(df.ensure(std::max<size_t>(n_out, 1) * 8)); MRGFE_TRY(dp.ensure(std::max<size_t>(n_out, 1) * 8));
    MRGFE_HIP_CHECK(hipMemcpyAsync(di.p, in, n_in * 8, hipMemcpyHostToDevice, ctx->stream));
    MRGFE_TRY(wave_fold_check_device(ctx, n_vals, di.as<double>(), cases, df.as<double>(), dp.as<double>()));
    MRGFE_HIP_CHECK(hipMemcpyAsync(out_fold, df.p, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipMemcpyAsync(out_plain, dp.p, n_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MRGFE_OK;
}

int mrgfe_dbg_exclusive_scan(mrgfe_ctx* ctx, const uint32_t* in, size_t n, uint32_t* out, uint32_t* total)
{
    MRGFE_TRY(check_count(n, "mrgfe_dbg_exclusive_scan"));
    if (!ctx || !total || (n && (!in || !out))) { set_error("mrgfe_dbg_exclusive_scan: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    uint32_t  nn = static_cast<uint32_t>(n);
    SliceTable tab;
    tab.build(&nn, 1);
    DevBuf &di = ctx->scratch[2], &dout = ctx->scratch[3], &db = ctx->scratch[8], &ds = ctx->scratch[0];
    MRGFE_TRY(di.ensure(std::max<size_t>(n, 1) * 4)); MRGFE_TRY(dout.ensure(std::max<size_t>(n, 1) * 4));
    MRGFE_TRY(db.ensure(sizeof(uint32_t) * (tab.total_blks + 8)));
    MRGFE_TRY(ds.ensure(sizeof(Slice)));
    MRGFE_HIP_CHECK(hipMemcpy(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice));
    if (n) MRGFE_HIP_CHECK(hipMemcpy(di.p, in, n * 4, hipMemcpyHostToDevice));
    uint32_t* d_tot = db.as<uint32_t>() + tab.total_blks;
    MRGFE_TRY(exclusive_scan(ctx, di.as<uint32_t>(), dout.as<uint32_t>(), ds.as<Slice>(), tab, db.as<uint32_t>(), d_tot));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n) MRGFE_HIP_CHECK(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    MRGFE_HIP_CHECK(hipMemcpy(total, d_tot, 4, hipMemcpyDeviceToHost));
    return MRGFE_OK;
}

int mrgfe_dbg_minmax(mrgfe_ctx* ctx, const float* xyzi, size_t n, float min3[3], float max3[3], uint32_t* n_finite)
{
    MRGFE_TRY(check_count(n, "mrgfe_dbg_minmax"));
    if (!ctx || !min3 || !max3 || !n_finite || (n && !xyzi)) { set_error("mrgfe_dbg_minmax: NULL argument"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    uint32_t  nn = static_cast<uint32_t>(n);
    SliceTable tab;
    tab.build(&nn, 1);
    DevBuf &dp = ctx->scratch[10], &dbb = ctx->scratch[1], &ds = ctx->scratch[0];
    MRGFE_TRY(dp.ensure(std::max<size_t>(n, 1) * 16));
    MRGFE_TRY(dbb.ensure(sizeof(BBox) * (tab.total_blks + 1)));
    MRGFE_TRY(ds.ensure(sizeof(Slice) + sizeof(void*)));
    MRGFE_TRY(upload_cloud(ctx, xyzi, n, 16, dp.p));
    const void* cp = dp.p;
    MRGFE_HIP_CHECK(hipMemcpy(ds.p, tab.h.data(), sizeof(Slice), hipMemcpyHostToDevice));
    MRGFE_HIP_CHECK(hipMemcpy(ds.as<char>() + sizeof(Slice), &cp, sizeof(void*), hipMemcpyHostToDevice));
    BBox* d_part = dbb.as<BBox>();
    BBox* d_out = d_part + tab.total_blks;
    MRGFE_TRY(bounding_boxes(ctx, reinterpret_cast<const float4* const*>(ds.as<char>() + sizeof(Slice)), ds.as<Slice>(), tab, d_part, d_out));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    BBox bb;
    MRGFE_HIP_CHECK(hipMemcpy(&bb, d_out, sizeof(BBox), hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; ++a) { min3[a] = bb.mn[a]; max3[a] = bb.mx[a]; }
    *n_finite = bb.n_finite;
    return MRGFE_OK;
}

// ---- the NDT optimiser state machine stepped by hand (no GPU involved): tests/test_controller_cpu.py feeds it the CPU oracle's
// derivative evaluations and must end where the oracle's own computeTransformation ends -------------------------------------
struct mrgfe_dbg_ctl { NdtController c; };

int mrgfe_dbg_set_host_control(int mode)
{
    ndt_set_host_control(mode);
    return MRGFE_OK;
}
int mrgfe_dbg_set_fused_launch(int mode) { return ndt_set_fused_launch(mode); }
int mrgfe_dbg_set_ndt_reference_order(int mode) { return ndt_set_reference_order(mode); }
int mrgfe_dbg_set_fit_sweep(int mode) { return nn_set_fit_sweep(mode); }
int mrgfe_dbg_set_fit_stats(int mode) { return nn_set_fit_stats(mode); }
void mrgfe_dbg_sincosf(const float* x, size_t n, float* sin_out, float* cos_out)
{
    for (size_t i = 0; i < n; ++i) { sin_out[i] = ctl::sin_f(x[i]); cos_out[i] = ctl::cos_f(x[i]); }
}
int mrgfe_dbg_exp(mrgfe_ctx* ctx, const double* x, size_t n, int on_device, double* out)
{
    if ((n && (!x || !out)) || (on_device && !ctx)) { set_error("mrgfe_dbg_exp: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!on_device) {
        for (size_t i = 0; i < n; ++i) out[i] = glibc_exp(x[i]);
        return MRGFE_OK;
    }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    DevBuf dx, dout;
    MRGFE_TRY(dx.ensure(std::max<size_t>(n, 1) * 8));
    int st = dout.ensure(std::max<size_t>(n, 1) * 8);
    if (st == MRGFE_OK && n) {
        if (hipMemcpyAsync(dx.p, x, n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) st = MRGFE_ERR_HIP;
        if (st == MRGFE_OK) st = glibc_exp_device(ctx, dx.as<double>(), n, dout.as<double>());
        if (st == MRGFE_OK && hipMemcpyAsync(out, dout.p, n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) st = MRGFE_ERR_HIP;
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && st == MRGFE_OK) st = MRGFE_ERR_HIP;
        if (st == MRGFE_ERR_HIP) set_error("mrgfe_dbg_exp: a HIP call failed");
    }
    dx.release();
    dout.release();
    return st;
}

int mrgfe_dbg_ctl_math(mrgfe_ctx* ctx, const double* cases48, int n, int on_device, float* M16, double* tables69, double* x6)
{
    if (!cases48 || !M16 || !tables69 || !x6 || n < 0) { set_error("mrgfe_dbg_ctl_math: bad argument"); return MRGFE_ERR_INVALID; }
    if (!on_device) {
        for (int i = 0; i < n; ++i) {
            const double* c = cases48 + size_t(i) * 48;
            ctl::pose_to_matrix(c, M16 + size_t(i) * 16);
            double j[8][3], h[15][3];
            ctl::angle_tables(c, j, h);
            std::memcpy(tables69 + size_t(i) * 69, j, sizeof(j));
            std::memcpy(tables69 + size_t(i) * 69 + 24, h, sizeof(h));
            ctl::svd_solve6(c + 6, c + 42, x6 + size_t(i) * 6);
        }
        return MRGFE_OK;
    }
    if (!ctx) { set_error("mrgfe_dbg_ctl_math: NULL context"); return MRGFE_ERR_INVALID; }
    MRGFE_LOCK(ctx);
    MRGFE_TRY(ctx->bind());
    DevBuf &din = ctx->scratch[0], &dM = ctx->scratch[1], &dt = ctx->scratch[2], &dx = ctx->scratch[3];
    const size_t nn = std::max(n, 1);
    MRGFE_TRY(din.ensure(nn * 48 * 8)); MRGFE_TRY(dM.ensure(nn * 64)); MRGFE_TRY(dt.ensure(nn * 69 * 8)); MRGFE_TRY(dx.ensure(nn * 48));
    MRGFE_HIP_CHECK(hipMemcpyAsync(din.p, cases48, size_t(n) * 48 * 8, hipMemcpyHostToDevice, ctx->stream));
    MRGFE_TRY(ndt_ctl_math_device(ctx, din.as<double>(), n, dM.as<float>(), dt.as<double>(), dx.as<double>()));
    if (on_device == 2) MRGFE_TRY(ndt_ctl_svd_wave_device(ctx, din.as<double>(), n, dx.as<double>()));  // x from the wavefront form of the solve
    MRGFE_HIP_CHECK(hipMemcpyAsync(M16, dM.p, size_t(n) * 64, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipMemcpyAsync(tables69, dt.p, size_t(n) * 69 * 8, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipMemcpyAsync(x6, dx.p, size_t(n) * 48, hipMemcpyDeviceToHost, ctx->stream));
    MRGFE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MRGFE_OK;
}
#ifdef MRGFE_TESTING
long mrgfe_dbg_fail_alloc_after(long k) { return fail_alloc_after(k); }
#endif

int mrgfe_batch_rounds(const mrgfe_batch* b) { return b && b->ndt ? b->ndt->rounds() : 0; }


int mrgfe_dbg_ctl_create(const mrgfe_reg_params* params, const float guess[16], uint32_t n_src, mrgfe_dbg_ctl** out)
{
    if (!params || !guess || !out) { set_error("mrgfe_dbg_ctl_create: NULL argument"); return MRGFE_ERR_INVALID; }
    if (!is_ndt(params->method)) { set_error("mrgfe_dbg_ctl_create: NDT_HIP / PCL_NDT_HIP only"); return MRGFE_ERR_INVALID; }
    MRGFE_TRY(check_params(params));
    mrgfe_dbg_ctl* h = new mrgfe_dbg_ctl();
    float g[16];
    col2row(guess, g);
    h->c.start(ndt_params_from(*params), g, n_src, std::getenv("MRGFE_DBG_CTL_SPLIT") != nullptr);
    if (ndt_set_reference_order(-1) && params->method == MRGFE_NDT_HIP) h->c.force_reference_solve();  // (as NdtEngine::align_all does in that mode)
    *out = h;
    return MRGFE_OK;
}
void mrgfe_dbg_ctl_destroy(mrgfe_dbg_ctl* h) { delete h; }
int mrgfe_dbg_ctl_request(const mrgfe_dbg_ctl* h, int* mode, float T[16], double p[6])
{
    if (!h || h->c.done()) return 0;
    const NdtCtlState& s = h->c.state();
    if (mode) *mode = s.req_mode;
    if (T) row2col(s.final_, T);
    if (p) std::memcpy(p, s.req_p, sizeof(double) * 6);
    return 1;
}
int mrgfe_dbg_ctl_result(mrgfe_dbg_ctl* h, double score, const double grad[6], const double hess[36], double neighbours)
{
    if (!h || !grad || !hess) { set_error("mrgfe_dbg_ctl_result: NULL argument"); return MRGFE_ERR_INVALID; }
    if (h->c.done()) { set_error("mrgfe_dbg_ctl_result: no request pending"); return MRGFE_ERR_STATE; }
    double r[kNdtPartialStride] = {0};
    r[0] = score;
    std::memcpy(r + 1, grad, sizeof(double) * 6);
    std::memcpy(r + 7, hess, sizeof(double) * 36);
    r[kNdtNbIndex] = neighbours;
    h->c.on_result(r);
    return MRGFE_OK;
}
int mrgfe_dbg_ctl_final(const mrgfe_dbg_ctl* h, float T[16], int* converged, int* iterations, int* evaluations)
{
    if (!h || !T) { set_error("mrgfe_dbg_ctl_final: NULL argument"); return MRGFE_ERR_INVALID; }
    row2col(h->c.final_transformation(), T);
    if (converged) *converged = h->c.converged() ? 1 : 0;
    if (iterations) *iterations = h->c.iterations();
    if (evaluations) *evaluations = h->c.evaluations();
    return MRGFE_OK;
}

}  // extern "C"

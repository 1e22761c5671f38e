/* include/mrgfe_debug.h — diagnostic and test entry points of libmrgfe.so, kept out of the integrator's header (include/mrgfe.h).
 *
 * None of these stands for a call of the reference.  `mrgfe_dbg_set_*` select the alternative path of the SAME arithmetic (host- or device-stepped optimiser,
 * fused or per-variant launches, …) so that tests hold the two against each other; the primitives (`sort_pairs`, `wave_sums`, `exclusive_scan`, `minmax`,
 * `grid_set_query`) expose building blocks to tests/test_gpu_primitives.py; `mrgfe_dbg_ctl_*` step the NDT optimiser by hand with no GPU involved
 * (oracle/replay.py, tests/test_controller_cpu.py).  They are exported by the shipped library.
 *
 * The two FAULT INJECTORS at the end are different: they make a correct call fail.  They are compiled only with -DMRGFE_TESTING — into
 * mrg_slam_amd/libmrgfe_testing.so, which tests/faultinject/ loads in a child process (MRGFE_LIB) — and the shipped libmrgfe.so does not contain them
 * (tests/test_abi.py checks both). */
#ifndef MRGFE_DEBUG_H
#define MRGFE_DEBUG_H
#include "mrgfe.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnostic entry points for the primitive tests (tests/test_gpu_primitives.py) --------------------------- */
/* correspondence search of GICP_HIP / SMALL_GICP_HIP (fast_gicp / small_gicp update_correspondences): the batched passes of getFitnessScore
 * carrying the index of the nearest point (csrc nn_nearest_batch) or one lane group per query until its answer is final.  1 = the passes for
 * batches of >= 400k queries (default: that is where they are faster), 0 = never, 2 = always.  Same correspondences either way; tests compare. */
int mrgfe_dbg_set_gicp_corr_passes(int mode);
/* Grids over `count` host clouds (packed xyzw floats) built TOGETHER (csrc NnGridSet, the batched form of the search grid under
 * getFitnessScore and the GICP covariances), then the `nq` queries answered against each: k == 1 the exact nearest neighbour, k > 1 the k
 * nearest, as mrgfe_knn.  idx / sqd: [count][nq][k].  `rounds` > 1 rebuilds the set that often (later builds reuse the cell edges). */
int mrgfe_dbg_grid_set_query(mrgfe_ctx* ctx, const float* const* clouds, const size_t* n, int count, const float* query, size_t nq, int k, int rounds, int32_t* idx, float* sqd);
int mrgfe_dbg_sort_pairs(mrgfe_ctx* ctx, const uint32_t* keys, const uint32_t* vals, size_t n, int key_bits, uint32_t* out_keys, uint32_t* out_vals);
/* The wave reduction of the derivative kernels' epilogue (csrc/dev_utils.h): in = cases x 64 lanes x n_vals doubles (n_vals 44, 37 or 1);
 * out_fold[cases][n_vals] from wave_sum_fold (n_vals values per lane folded in six steps), out_plain from n_vals separate wave_sum
 * calls.  Same summation tree, so the two must agree bit for bit (tests/test_gpu_primitives.py). */
int mrgfe_dbg_wave_sums(mrgfe_ctx* ctx, int n_vals, const double* in, int cases, double* out_fold, double* out_plain);
int mrgfe_dbg_exclusive_scan(mrgfe_ctx* ctx, const uint32_t* in, size_t n, uint32_t* out, uint32_t* total);
int mrgfe_dbg_minmax(mrgfe_ctx* ctx, const float* xyzi, size_t n, float min3[3], float max3[3], uint32_t* n_finite);

/* The NDT optimiser (Newton + More-Thuente state machine: pclomp computeTransformation / computeStepLengthMT, the same source
 * the device steps in ndt_reduce_kernel) driven by hand, no GPU involved: create() runs align()'s prologue for `guess`;
 * request() returns 1 and the pending derivative evaluation (mode 0 score+gradient+Hessian, 1 score+gradient, 2 f64 Hessian
 * only; T = the transform to evaluate at, column-major; p = its pose vector) or 0 when the alignment is finished; result() hands
 * that evaluation's sums over and advances to the next request.  tests/test_controller_cpu.py feeds it the CPU oracle's
 * evaluations. */
/* Who steps that optimiser during the following alignments of this process: 0 = the device (the batch advances round after round
 * without the host), 1 = the host (one synchronisation per round), -1 = automatic (default: single registrations on the host,
 * batches on the device; the environment variable MRGFE_HOST_CONTROL sets the initial value).  Results are the same either way;
 * tests/test_gpu_control.py holds the two against each other. */
int mrgfe_dbg_set_host_control(int mode);
/* How the derivative evaluations of a round are launched during the following alignments of this process: 1 = ONE launch for all
 * three kernel variants, their work items walked kind after kind (default; the environment variable MRGFE_FUSED sets the initial value),
 * 0 = one launch per variant.  Any other value only asks.  Returns the setting in effect.  Same sums either way: an item's partial
 * record does not depend on the launch it is computed in (tests/test_gpu_control.py). */
int mrgfe_dbg_set_fused_launch(int mode);
/* NDT_HIP's f64 sums during the following alignments of this process: 1 = in the REFERENCE's order — ndt_omp adds a point's voxel terms from zero, then the
 * per-point sums point after point ("invariant against the summing up order"), and computeHessian pair after pair on one thread — reproduced by a record
 * kernel + one dependent-add chain per accumulator (csrc/ndt_derivatives.hip ndt_ref_*): bit-identical to the reference-order oracle, results inside the
 * 1e-4 bar unconditionally, several times slower (a 130k-step chain per evaluation whatever the batch size; host-stepped rounds); 0 = the tree (default; the
 * environment variable MRGFE_NDT_REFERENCE_ORDER sets the initial value).  Any other value only asks.  Returns the setting in effect. */
int mrgfe_dbg_set_ndt_reference_order(int mode);
/* How getFitnessScore's far pass runs during the following calls of this process: 1 = seed + sweep (nn_fit_sweep_kernel: a near occupied
 * cell found through the occupancy words gives a radius, the occupied cells inside it are enumerated top-down with bit masks; default,
 * MRGFE_FIT_SWEEP sets the initial value), 0 = round 2's pyramid walk for every queued query.  Any other value only asks.  Returns the
 * setting in effect.  Both give the exact nearest distances (tests/test_gpu_fitness_passes.py). */
int mrgfe_dbg_set_fit_sweep(int mode);
/* mrgfe_prefilter / mrgfe_prefilter_device with VoxelGrid + RadiusOutlierRemoval: 1 (default) the stages' point counts stay on the device and the
 * call waits once at its end, 0 every stage reports its count to the host (round 3; also what an unusual scan falls back to); other values
 * query.  Same outputs either way. */
int mrgfe_dbg_set_prefilter_device_driven(int mode);
/* PCL_GICP_HIP (serial pcl::GeneralizedIterativeClosestPoint, registrations.cpp:93-103): 1 (default) the thirteen sums of every cost / gradient
 * evaluation are added in the reference's order, point after point (bit-identical BFGS trajectories; ~4 ns per point and evaluation), 0 in a tree
 * (round 3: faster, and outside the 1e-4 bar on one random scene in fourteen); other values query.  PCL_GICP_OMP_HIP: 1 = pclomp's per-thread chunk
 * sums for its stated thread count (n / T dependent additions per evaluation), 0 = the tree. */
int mrgfe_dbg_set_pclgicp_reference_order(int mode);
/* Counters of the seed + sweep pass (mrgfe_ctx_fitness_stats out[6..9], mrgfe_batch_fitness_stats) during the following calls of this process:
 * 0 = off (default; MRGFE_FIT_STATS sets the initial value), 1 = counted, 2 = also the kernel's phase clocks and a line on stderr (slows
 * the kernel: a clock read waits for the memory operations in flight).  Any other value only asks.  Returns the setting in effect. */
int mrgfe_dbg_set_fit_stats(int mode);
/* The optimiser's scalar routines (pose vector -> float matrix, angle derivative tables, 6x6 SVD solve) on n cases of 48 doubles
 * (p[6], A[36] row-major, b[6]), on the host (on_device = 0, ctx may be NULL) or on the device: M16 [n][16] row-major, tables69
 * [n][8*3 + 15*3], x6 [n][6]; on_device = 2: x6 from the wavefront form of the solve (three lanes rotate, 36 apply) that the
 * device controller uses.  One source (csrc/ndt_ctl.h) compiled twice; the test compares the builds bit for bit. */
/* the float sine / cosine the optimiser builds its pose matrices with (host build of csrc/ndt_ctl.h: glibc's sinf / cosf algorithm
 * restated, because the reference's Eigen::AngleAxisf calls exactly those and they are not correctly rounded) */
void mrgfe_dbg_sincosf(const float* x, size_t n, float* sin_out, float* cos_out);
/* glibc's double exp() as the kernels compute it (csrc/glibc_exp.h: the reference evaluates its per-pair weights with the host's libm, and exp is not
 * correctly rounded — the device library's differs in the last bit on one argument in ten): on the host (on_device = 0, ctx may be NULL) or on the device.
 * tests/test_glibc_exp.py holds the host build against the C library and regenerates the table; tests/test_gpu_primitives.py holds the device against the host. */
int mrgfe_dbg_exp(mrgfe_ctx* ctx, const double* x, size_t n, int on_device, double* out);
int mrgfe_dbg_ctl_math(mrgfe_ctx* ctx, const double* cases48, int n, int on_device, float* M16, double* tables69, double* x6);
typedef struct mrgfe_dbg_ctl mrgfe_dbg_ctl;
int  mrgfe_dbg_ctl_create(const mrgfe_reg_params* params, const float guess[16], uint32_t n_src, mrgfe_dbg_ctl** out);
void mrgfe_dbg_ctl_destroy(mrgfe_dbg_ctl* h);
int  mrgfe_dbg_ctl_request(const mrgfe_dbg_ctl* h, int* mode, float T[16], double p[6]);
int  mrgfe_dbg_ctl_result(mrgfe_dbg_ctl* h, double score, const double grad[6], const double hess[36], double neighbours);
int  mrgfe_dbg_ctl_final(const mrgfe_dbg_ctl* h, float T[16], int* converged, int* iterations, int* evaluations);

#ifdef MRGFE_TESTING
/* hardening hook: the k-th device / pinned allocation of this process from now on (0 = the next one) fails as if the device were out of memory, every
 * later one works again; k < 0 switches the injector off (MRGFE_FAIL_ALLOC_AFTER sets the initial value).  Returns the number of allocations made
 * since the previous call: a test sweeps k over a whole entry point and wants an error code from every k, then a correct answer. */
long   mrgfe_dbg_fail_alloc_after(long k);
/* test hook: the next mrgfe_node_align fails on that member (the error path without an out-of-memory condition) */
int    mrgfe_dbg_node_fail_member(mrgfe_node* node, int member);
#endif /* MRGFE_TESTING */

#ifdef __cplusplus
}
#endif
#endif /* MRGFE_DEBUG_H */

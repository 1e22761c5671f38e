/* include/mrgfe.h — C ABI of libmrgfe.so, the MI355X (gfx950) scan-matching front end for mrg_slam.
 *
 * Drop-in boundary: everything the reference reaches through
 *     pcl::Registration<PointXYZI,PointXYZI>::Ptr select_registration_method(rclcpp::Node*)
 *         /root/reference/include/mrg_slam/registrations.hpp:20, src/mrg_slam/registrations.cpp:28-152
 * and through the pcl::Filter objects held by the prefiltering component
 *         /root/reference/apps/prefiltering_component.cpp:158-229
 * is exported here as plain C: opaque handles, host (or device) pointers and sizes, int status codes.
 * No exceptions cross this boundary; no torch / PCL / Eigen types appear in it.  The existing precedent for a
 * GPU registration back end in the reference is the FAST_VGICP_CUDA branch (registrations.cpp:65-75,
 * CMakeLists.txt:45-49); include/mrgfe_pcl_adapter.hpp wraps this ABI into that same slot.
 *
 * Conventions
 *   - clouds: float x,y,z,intensity per point ("xyzi").  Every `stride_bytes` argument is a point-layout descriptor:
 *     16 (or 0) = packed x,y,z,intensity records (KITTI .bin, the replay scripts' PointCloud2, every output of this
 *     library); any other record layout is named with MRGFE_LAYOUT(stride, xyz_offset, intensity_offset) —
 *     MRGFE_LAYOUT_PCL_XYZI for the reference's in-memory pcl::PointXYZI (32 bytes: x,y,z at 0, a 1.0f padding word at 12,
 *     intensity at byte 16).  A bare stride other than 16 is refused (MRGFE_ERR_INVALID): it cannot say where the
 *     intensity lives.  Strided records go to the device as they are and are gathered there.  Inputs are copied to the
 *     device inside the call: the caller may free or reuse its buffer when the call returns.
 *   - 4x4 matrices: float[16] / double[16], COLUMN-major (Eigen's default, i.e. Eigen::Matrix4f::data()).
 *   - 6x6 matrices: double[36], row-major (symmetric in practice).
 *   - status: 0 = ok, <0 = error; mrgfe_last_error() returns a thread-local message for the last failure.
 *     Non-convergence is NOT an error (pcl::Registration::align returns void; callers test hasConverged():
 *     apps/scan_matching_odometry_component.cpp:270, src/mrg_slam/loop_detector.cpp:138).
 *   - a handle is used from one thread at a time; distinct handles may be used from different threads (the odometry
 *     and loop-closure registrations of the reference live in different threads): calls that share a context are
 *     serialised by a per-context lock, calls on different contexts (e.g. one per GPU) run concurrently.
 */
#ifndef MRGFE_H
#define MRGFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* point-layout descriptor for the `stride_bytes` arguments: stride (<= 65532) | (intensity byte offset + 1) << 16 | (byte offset of
 * x; y and z follow it) << 24; offsets < 252, multiples of 4 */
#define MRGFE_LAYOUT(stride, xyz_offset, intensity_offset) \
    ((size_t)(stride) | ((size_t)((intensity_offset) + 1) << 16) | ((size_t)(xyz_offset) << 24))
#define MRGFE_LAYOUT_PACKED ((size_t)16)
#define MRGFE_LAYOUT_PCL_XYZI MRGFE_LAYOUT(32, 0, 16) /* sizeof(pcl::PointXYZI) == 32, intensity behind the padded xyz quad */

#define MRGFE_OK 0
#define MRGFE_ERR_INVALID (-1)   /* bad argument / handle */
#define MRGFE_ERR_HIP (-2)       /* HIP runtime error (no GPU, out of memory, launch failure) */
#define MRGFE_ERR_OVERFLOW (-3)  /* voxel index would overflow int32 ("Leaf size is too small for the input dataset") */
#define MRGFE_ERR_EMPTY (-4)     /* operation needs a non-empty cloud */
#define MRGFE_ERR_STATE (-5)     /* target / source not set */

typedef struct mrgfe_ctx mrgfe_ctx; /* one per (process, GPU): stream + grow-only device workspace */
typedef struct mrgfe_reg mrgfe_reg; /* one registration object == one pcl::Registration instance */

/* registration_method strings of the reference that this library serves (registrations.cpp:45-151) */
enum mrgfe_method {
    MRGFE_NDT_HIP = 0,  /* replaces "NDT_OMP"  : pclomp::NormalDistributionsTransform (registrations.cpp:130-148) */
    MRGFE_GICP_HIP = 1, /* replaces "FAST_GICP": fast_gicp::FastGICP                  (registrations.cpp:55-63)   */
    MRGFE_SMALL_GICP_HIP = 2, /* replaces "SMALL_GICP" (the YAML default, config/mrg_slam.yaml:100): small_gicp::RegistrationPCL
                                 (registrations.cpp:46-54): the same GICP factor perturbed on the right, small_gicp's LM schedule */
    MRGFE_VGICP_HIP = 3, /* replaces "FAST_VGICP" (fast_gicp::FastVGICP, registrations.cpp:76-84) and the reference's own GPU slot
                            "FAST_VGICP_CUDA" (:65-75): voxelised GICP, target as a Gaussian voxel map of edge `resolution` */
    MRGFE_ICP_HIP = 4, /* replaces "ICP": pcl::IterativeClosestPoint (registrations.cpp:85-92), use_reciprocal_correspondences as in :91;
                          single registrations only (not in mrgfe_batch_*) */
    MRGFE_PCL_GICP_HIP = 5, /* replaces "GICP": pcl::GeneralizedIterativeClosestPoint (registrations.cpp:93-103): PCL's covariances, nearest-point
                               correspondences, inner BFGS with max_optimizer_iterations steps; single registrations only */
    MRGFE_PCL_GICP_OMP_HIP = 6, /* replaces "GICP_OMP": pclomp::GeneralizedIterativeClosestPoint (registrations.cpp:104-114): the same algorithm with
                                   the older stopping rule of the inner BFGS (norm of the whole gradient < 1e-2) and pclomp's accumulation: per-thread
                                   partial sums over static chunks of the correspondences, added in thread order — for num_threads threads (0: 8; at
                                   most 16).  Upstream the count is omp_get_max_threads() of the host: the reference's result depends on it */
    MRGFE_PCL_NDT_HIP = 7 /* replaces "NDT" and every name the factory does not know: pcl::NormalDistributionsTransform (registrations.cpp:115-129;
                             PCL 1.12): f64 pair terms, the radius search over the voxel centroids (nn_search_method is ignored), PCL's own
                             iteration test (squared translation of the last step <= transformation_epsilon); also in mrgfe_batch_* */
};
/* reg_nn_search_method (registrations.cpp:140-146) */
enum mrgfe_ndt_search { MRGFE_KDTREE = 0, MRGFE_DIRECT26 = 1, MRGFE_DIRECT7 = 2, MRGFE_DIRECT1 = 3 };

/* Mirrors the ten reg_* ROS parameters read by select_registration_method (registrations.cpp:34-43) plus the
 * library defaults those classes carry (ndt_omp: step_size 0.1, outlier_ratio 0.55; fast_gicp: rotation_epsilon 2e-3). */
typedef struct mrgfe_reg_params {
    int    method;                          /* enum mrgfe_method                        "registration_method"               */
    int    num_threads;                     /* unused on the GPU, except PCL_GICP_OMP_HIP: "reg_num_threads"
                                               the OpenMP thread count whose order of additions is reproduced: 1..16
                                               (0 = 8, the YAML's value; 1 = one chain; more than 16 is refused)            */
    double transformation_epsilon;          /*                                          "reg_transformation_epsilon"        */
    int    maximum_iterations;              /*                                          "reg_maximum_iterations"            */
    double max_correspondence_distance;     /* GICP                                     "reg_max_correspondence_distance"   */
    int    max_optimizer_iterations;        /* PCL_GICP_HIP: steps of the inner BFGS    "reg_max_optimizer_iterations"      */
    int    use_reciprocal_correspondences;  /* ICP_HIP (pcl::GICP ignores it upstream)  "reg_use_reciprocal_correspondences"*/
    int    correspondence_randomness;       /* GICP k neighbours                        "reg_correspondence_randomness"     */
    double resolution;                      /* NDT / VGICP voxel size                   "reg_resolution"                    */
    int    nn_search_method;                /* enum mrgfe_ndt_search                    "reg_nn_search_method"              */
    double step_size;                       /* NDT More-Thuente step_max (0.1)    */
    double outlier_ratio;                   /* NDT (0.55)                         */
    double rotation_epsilon;                /* GICP (2e-3)                        */
} mrgfe_reg_params;

/* ---- library / context ------------------------------------------------------------------------------------- */
const char* mrgfe_last_error(void);
const char* mrgfe_version(void);
int  mrgfe_ctx_create(int device_id, mrgfe_ctx** out);
/* the same with the context's stream at the device's highest stream priority (high_priority != 0): workgroups of its kernels are
 * dispatched ahead of those of normal contexts as slots free up — for the latency-critical odometry registration of a process that
 * also runs loop-closure batches (scan_matching_odometry_component next to mrg_slam_component's LoopDetector) */
int  mrgfe_ctx_create_priority(int device_id, int high_priority, mrgfe_ctx** out);
/* a context whose kernels never occupy `reserve_cus` of the device's compute units (HIP stream with a CU mask; its helper streams too): for the
 * throughput work of a process — loop-closure batches (mrg_slam_component's LoopDetector) — so that the small per-scan launches of the odometry
 * contexts beside it always find free compute units instead of waiting for a batch's workgroups to drain.  reserve_cus = 0: mrgfe_ctx_create;
 * MRGFE_RESERVE_AUTO: sized from the device; reserve_cus >= the device's compute units: MRGFE_ERR_INVALID. */
#define MRGFE_RESERVE_AUTO (-1) /* reserve_cus: a quarter of the device's compute units, at most 64 (MI355X: 64 of 256; a 32-CU partition: 8) */
int  mrgfe_ctx_create_reserving(int device_id, int reserve_cus, mrgfe_ctx** out);
/* (HIP creates compute-unit-masked streams with default flags: unlike every other stream of this library they synchronise with the legacy NULL stream.  A
 * process that also enqueues work on the NULL stream — torch's default stream, a synchronous hipMemcpy — serialises a reserving context's kernels
 * against it; keep NULL-stream work out of such a process or use per-thread default streams.  Results never depend on it.) */
void mrgfe_ctx_destroy(mrgfe_ctx* ctx);
int  mrgfe_ctx_synchronize(mrgfe_ctx* ctx);
/* page-lock a host buffer the caller keeps between calls (hipHostRegister): downloads into it and uploads out of it are direct DMA at PCIe rate
 * instead of staged through the runtime's bounce buffers — worth it for the 60 MB map cloud of mrgfe_map_store_generate / mrgfe_map_cloud_generate
 * (map_cloud_generator.cpp:14-86 returns a fresh pcl cloud per call; a caller of this library reuses one buffer).  Unpin before freeing it. */
int  mrgfe_pin_host_buffer(mrgfe_ctx* ctx, void* p, size_t bytes);
int  mrgfe_unpin_host_buffer(mrgfe_ctx* ctx, void* p);
/* Zero-copy uploads (round 5; off by default).  on = 1: a packed cloud (stride 16) of 64 KB or more that is handed to mrgfe_*_set_* / mrgfe_batch_add_* /
 * mrgfe_node_add_* on this context out of page-locked host memory is read by DMA straight from the caller's buffer instead of through the context's
 * staging ring (a single-thread copy at about half the link's rate: 256 config[1] pairs from host clouds 44 -> 34.5 ms per step, 28.4 ms with two
 * batches in flight = 37 GB/s over the link).  The DMA is stream-ordered, so the CONTRACT CHANGES for such buffers: keep them unchanged until the call
 * that consumes the cloud (…_align / …_wait, mrgfe_ctx_synchronize) has returned — the reference's keyframe clouds are immutable ConstPtr clouds.
 * Pageable buffers, and everything while the switch is off, keep the default contract: free or reuse the memory as soon as the add / set call returns.
 * (mrgfe_node_* declares its clouds by pointer and uploads them inside mrgfe_node_align: its members always run with the switch on.) */
int  mrgfe_ctx_set_zero_copy_uploads(mrgfe_ctx* ctx, int on);
/* HIP stream of the context as an opaque pointer (hipStream_t), for callers that order their own work after it */
void* mrgfe_ctx_stream(mrgfe_ctx* ctx);
/* Measurement hook ("1-NN fitness on hash grid: N (16 + 27*8 + m*16)"): what the last getFitnessScore pass on this context
 * did — mrgfe_reg_fitness, mrgfe_batch_align(fitness_max_range >= 0), mrgfe_calc_fitness_score, mrgfe_map_store_fitness.
 * out[0..2] = HIP-event milliseconds of the block pass / the seed + sweep pass / the pyramid walk of what could not be seeded (events on
 * the context's stream), out[3] = queries, out[4] = queries their 3x3x3 block did not settle, out[5] = queries without a seed within three
 * blocks, out[6..9] = occupancy words fetched / boxes tested against the sphere / cells opened / candidate points measured by the seed +
 * sweep pass (0 unless MRGFE_FIT_STATS=1), out[10] = number of passes run on this context so far. */
int mrgfe_ctx_fitness_stats(mrgfe_ctx* ctx, double out[11]);
/* The same for the last exact k-NN launch on this context (nn_knn_kernel: the k = 20 neighbourhoods behind the GICP covariances of
 * setInputSource / setInputTarget, registrations.cpp:46-63; StatisticalOutlierRemoval; mrgfe_knn): out[0] = HIP-event milliseconds of the
 * launch, out[1] = queries, out[2] = k, out[3] = candidate points measured (0 unless the diagnostic counters are on,
 * mrgfe_dbg_set_fit_stats), out[4] = k-NN launches on this context so far.  Waits for that launch to finish. */
int mrgfe_ctx_knn_stats(mrgfe_ctx* ctx, double out[5]);

/* ---- cloud ingest --------------------------------------------------------------------------------- */
/* replaces pcl::fromROSMsg(*cloud_msg, *cloud) (apps/prefiltering_component.cpp:119-120, scan_matching_odometry_component.cpp:144-145):
 * the byte payload of a sensor_msgs/PointCloud2 (little-endian FLOAT32 fields at the given byte offsets; row_step 0 means
 * width * point_step; off_intensity < 0: no such field, intensity 0 like PointXYZI's default) becomes a packed x,y,z,intensity
 * cloud in host memory (out_xyzi, width*height*16 bytes, may be NULL) and / or device memory (d_out_xyzi, may be NULL) — the
 * latter feeds mrgfe_reg_set_*_device / mrgfe_batch_add_*_device / mrgfe_prefilter_device without the cloud leaving HBM.
 * The replay scripts' layout (point_step 16, offsets 0/4/8/12: python_scripts/kitti_singlerobot_processor.py:164-185) is a plain
 * copy; anything else is gathered on the device.  `data` must hold (height - 1) * row_step + width * point_step bytes: there is no
 * length argument (a sensor_msgs/PointCloud2 guarantees it; the Python wrapper checks it). */
int mrgfe_ingest_pointcloud2(mrgfe_ctx* ctx, const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step, uint32_t row_step, uint32_t off_x, uint32_t off_y,
                             uint32_t off_z, int32_t off_intensity, float* out_xyzi, void* d_out_xyzi);

/* ---- registration: the pcl::Registration call surface the reference uses (SURVEY.md §8b "Seam") ---------------- */
/* defaults of select_registration_method's parameters (registrations.cpp:34-43 comments) for `method` */
void mrgfe_reg_default_params(int method, mrgfe_reg_params* out);
/* replaces: new pclomp::NormalDistributionsTransform + setters (registrations.cpp:133-146) / new fast_gicp::FastGICP (:57-62) */
int  mrgfe_reg_create(mrgfe_ctx* ctx, const mrgfe_reg_params* params, mrgfe_reg** out);
void mrgfe_reg_destroy(mrgfe_reg* reg);
/* replaces registration_->setInputTarget(cloud): scan_matching_odometry_component.cpp:203,295,333; loop_detector.cpp:104.
 * NDT: builds the voxel covariance grid (pclomp::VoxelGridCovariance::filter). Returns MRGFE_ERR_OVERFLOW like PCL's
 * index-overflow abort (the registration then has no target). */
int  mrgfe_reg_set_target(mrgfe_reg* reg, const float* xyzi, size_t n, size_t stride_bytes);
/* replaces registration_->setInputSource(cloud): scan_matching_odometry_component.cpp:208; loop_detector.cpp:127,229,272 */
int  mrgfe_reg_set_source(mrgfe_reg* reg, const float* xyzi, size_t n, size_t stride_bytes);
/* same, from packed float4 clouds already resident in device memory (zero-copy ingest; bench.py's timed region) */
int  mrgfe_reg_set_target_device(mrgfe_reg* reg, const void* d_xyzi, size_t n);
int  mrgfe_reg_set_source_device(mrgfe_reg* reg, const void* d_xyzi, size_t n);
/* mrgfe_reg_set_source_device for the cloud the last mrgfe_prefilter_device call on this registration's context left in `d_xyzi` (n = the count it
 * returned), UNTOUCHED since: the prefilter chain knows a box that encloses its output, so the GICP family builds the source's search grid without its own
 * bounding-box pass and stream wait.  Anything else (another pointer or count, a cloud that went to the host, NDT) is mrgfe_reg_set_source_device. */
int  mrgfe_reg_set_source_from_prefilter(mrgfe_reg* reg, const void* d_xyzi, size_t n);
/* replaces registration_->setInputTarget(keyframe) where the keyframe IS the cloud last given to set_source (the odometry's keyframe update,
 * scan_matching_odometry_component.cpp:326-339 -> :333): same result as set_target of that cloud, without uploading it again and, for the GICP
 * family, without recomputing its k-NN covariances and search grid (they were made when it was the source).  The source stays set. */
int  mrgfe_reg_source_becomes_target(mrgfe_reg* reg);
/* replaces registration_->align(*aligned, guess): scan_matching_odometry_component.cpp:265-266; loop_detector.cpp:134,236,279.
 * `aligned_xyzi` (n_source packed float4, may be NULL) receives final_transformation * source. */
int  mrgfe_reg_align(mrgfe_reg* reg, const float guess[16], float* aligned_xyzi);
/* replaces hasConverged() / getFinalTransformation(): scan_matching_odometry_component.cpp:270,275; loop_detector.cpp:138,144 */
int  mrgfe_reg_has_converged(const mrgfe_reg* reg);
int  mrgfe_reg_final_transformation(const mrgfe_reg* reg, float out[16]);
/* replaces getFitnessScore(max_range): loop_detector.cpp:137; scan_matching_odometry_component.cpp:403.
 * PCL semantics: mean squared 1-NN distance over source points whose SQUARED distance is <= max_range. */
int  mrgfe_reg_fitness(mrgfe_reg* reg, double max_range, double* out);
/* replaces getSearchMethodTarget()->nearestKSearch(pt, 1, ..): scan_matching_odometry_component.cpp:405-417 (batched) */
int  mrgfe_reg_nn1_target(mrgfe_reg* reg, const float* query_xyzi, size_t n, size_t stride_bytes, int32_t* idx, float* sqdist);
/* extra read-outs (pcl: getFinalNumIteration / ndt: getTransformationProbability; 6x6 Hessian for the pose gather of §8e) */
int    mrgfe_reg_iterations(const mrgfe_reg* reg);
int    mrgfe_reg_evaluations(const mrgfe_reg* reg); /* derivative (NDT) / linearize+error (GICP) passes of the last align */
double mrgfe_reg_trans_probability(const mrgfe_reg* reg);
int    mrgfe_reg_hessian(const mrgfe_reg* reg, double out[36]);

/* ---- NDT internals exposed for kernel-level parity tests and the roofline accounting --------------------------- */
/* one derivative evaluation at pose vector p (tx,ty,tz,rx,ry,rz) with the source transformed by T:
 * mode 0 = score+gradient+Hessian, 1 = score+gradient, 2 = Hessian only in double (pclomp computeHessian).  PCL_NDT_HIP: all three in f64. */
int mrgfe_ndt_evaluate(mrgfe_reg* reg, const float T[16], const double p[6], int mode, double* score, double grad[6], double hess[36]);
/* target grid: number of voxels with >= 1 point; per-voxel key (ascending), point count (-1: rejected by the
 * eigenvalue / inf checks), mean[3], inverse covariance[9] (row-major). Arrays sized by mrgfe_ndt_num_leaves. */
int mrgfe_ndt_num_leaves(const mrgfe_reg* reg);
int mrgfe_ndt_grid(const mrgfe_reg* reg, int32_t min_b[3], int32_t max_b[3], int32_t div_b[3]);
int mrgfe_ndt_leaves(mrgfe_reg* reg, int32_t* keys, int32_t* nr_points, double* mean3, double* icov9);
/* mean number of valid neighbour voxels per source point over the evaluations of the last align (k-bar of SURVEY §8d) */
double mrgfe_ndt_mean_neighbours(const mrgfe_reg* reg);

/* k nearest neighbours of every query among the points of `cloud` (pcl::search::KdTree::nearestKSearch(pt, k, ...), the
 * search behind fast_gicp's covariances and StatisticalOutlierRemoval): idx / sqd are [nq][k], ascending by (squared
 * distance, index); entries beyond the cloud's size are -1 / -1.  1 <= k <= 64.  Non-finite queries get no neighbours. */
int mrgfe_knn(mrgfe_ctx* ctx, const float* cloud_xyzi, size_t n, const float* query_xyzi, size_t nq, size_t stride_bytes, int k, int32_t* idx, float* sqd);

/* ---- GICP internals exposed for kernel-level parity tests ------------------------------------------------------------ */
/* update_correspondences + linearize at the pose T (column-major double 4x4, T_target_source): H (row-major 6x6,
 * rotation block first), b, the sum of r^T M r over the correspondences (the variants scale it themselves) and their
 * number.  The Jacobian is the one of the registration's method (fast_gicp: left, small_gicp: right perturbation). */
int mrgfe_gicp_linearize(mrgfe_reg* reg, const double T[16], double H[36], double b[6], double* sum_errors, int* n_correspondences);
/* regularised k-NN covariances (row-major 3x3 per point) of the source (which = 0) or target (1) cloud */
/* PCL_GICP_HIP (tests): the correspondences and Mahalanobis matrices pcl::GICP's search loop finds at transformation_ = T (column-major, guess
 * = identity), and the cost estimateRigidTransformationBFGS minimises at x = (t, euler ZYX) over them: *f, grad[6], number of correspondences */
int mrgfe_pclgicp_evaluate(mrgfe_reg* reg, const float T[16], const double x[6], double* f, double grad[6], int* n_correspondences);
int mrgfe_gicp_covariances(mrgfe_reg* reg, int which, double* cov9_per_point);

/* ---- prefilter chain (apps/prefiltering_component.cpp:149-151). Outputs: caller-allocated capacity-n packed float4
 *      buffers + count.  Order-preserving where the reference is. ------------------------------------------------- */
/* The whole chain of PrefilteringComponent::cloud_callback (:149-151: distance_filter -> downsample -> outlier_removal) in
 * one call: the cloud goes up once, stays in HBM between the three passes, and comes down once.  Same output as the three
 * calls below one after the other.  Parameter names / YAML defaults: config/mrg_slam.yaml:41-64
 * (prefiltering_component.cpp:92-112 declares them). */
typedef struct mrgfe_prefilter_params {
    int    enable_distance_filter;            /* 1                                     */
    double distance_near_thresh;              /* 0.1                                   */
    double distance_far_thresh;               /* 35.0                                  */
    int    downsample_method;                 /* 0 NONE, 1 VOXELGRID, 2 APPROX_VOXELGRID */
    double downsample_resolution;             /* 0.1                                   */
    int    downsample_min_points_per_voxel;   /* 1                                     */
    int    outlier_removal_method;            /* 0 NONE, 1 RADIUS, 2 STATISTICAL       */
    double radius_radius;                     /* 0.5                                   */
    int    radius_min_neighbors;              /* 2                                     */
    int    statistical_mean_k;                /* 30                                    */
    double statistical_stddev;                /* 1.2                                   */
} mrgfe_prefilter_params;
void mrgfe_prefilter_default_params(mrgfe_prefilter_params* out);
int  mrgfe_prefilter(mrgfe_ctx* ctx, const mrgfe_prefilter_params* params, const float* xyzi, size_t n, size_t stride_bytes, float* out_xyzi, size_t* out_n);
/* the same with the result left in device memory (packed float4, capacity >= n points) for mrgfe_reg_set_source_device /
 * mrgfe_batch_add_*_device: the filtered scan goes from the prefiltering callback to the scan matcher without leaving HBM */
int  mrgfe_prefilter_device(mrgfe_ctx* ctx, const mrgfe_prefilter_params* params, const float* xyzi, size_t n, size_t stride_bytes, void* d_out_xyzi, size_t* out_n);

/* replaces PrefilteringComponent::distance_filter (:206-229): keep iff near < |p| < far */
int mrgfe_distance_filter(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, double near_thresh, double far_thresh,
                          float* out_xyzi, size_t* out_n);
/* replaces pcl::VoxelGrid<PointXYZI>::filter with setLeafSize(l,l,l), setMinimumPointsNumberPerVoxel (:168-171;
 * scan_matching_odometry_component.cpp:176-179). Returns MRGFE_OK with *overflow=1 and output == input when PCL would
 * warn "Leaf size is too small" and pass the cloud through. */
int mrgfe_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, float leaf, int min_points_per_voxel,
                    float* out_xyzi, size_t* out_n, int* overflow);
/* replaces pcl::ApproximateVoxelGrid<PointXYZI>::filter (downsample_method APPROX_VOXELGRID: prefiltering_component.cpp:172-175,
 * scan_matching_odometry_component.cpp:180-183): the 512-entry direct-mapped history of cells (hash (ix*7171 + iy*3079 + iz*4231) & 511), an entry
 * flushed — its float centroid of x, y, z, intensity emitted — whenever a point of another cell arrives at it, the rest flushed in entry order at
 * the end.  The output depends on the order of the input, a cell can be emitted several times, and there is no minimum point count: all of
 * that is reproduced point for point (csrc/filters.hip decomposes the sequential loop into the 512 independent per-entry sequences).
 * out_xyzi: capacity n points. */
int mrgfe_approx_voxelgrid(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, float leaf, float* out_xyzi, size_t* out_n);
/* replaces pcl::RadiusOutlierRemoval::filter with setRadiusSearch / setMinNeighborsInRadius (:195-198) */
int mrgfe_radius_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, double radius, int min_neighbors,
                         float* out_xyzi, size_t* out_n);
/* replaces pcl::StatisticalOutlierRemoval::filter with setMeanK / setStddevMulThresh (:189-192) */
int mrgfe_statistical_outlier(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, int mean_k, double stddev_mul,
                              float* out_xyzi, size_t* out_n);
/* replaces InformationMatrixCalculator::calc_fitness_score (src/mrg_slam/information_matrix_calculator.cpp:46-81) */
int mrgfe_calc_fitness_score(mrgfe_ctx* ctx, const float* cloud1, size_t n1, const float* cloud2, size_t n2, size_t stride_bytes,
                             const double relpose[16], double max_range, double* out);

/* InformationMatrixCalculator's parameters (apps/mrg_slam_component.cpp:313-322 declares them; defaults: config/mrg_slam.yaml:216-223,173) */
typedef struct mrgfe_inf_params {
    int    use_const_inf_matrix;          /* 0    */
    double const_stddev_x, const_stddev_q; /* 0.5, 0.1 */
    double var_gain_a;                    /* 2.0  */
    double min_stddev_x, max_stddev_x;    /* 0.1, 0.75 */
    double min_stddev_q, max_stddev_q;    /* 0.05, 0.2 */
    double fitness_score_thresh;          /* 1.25 */
} mrgfe_inf_params;
void   mrgfe_inf_default_params(mrgfe_inf_params* out);
/* replaces InformationMatrixCalculator::weight (information_matrix_calculator.cpp:83-88) */
double mrgfe_inf_weight(double a, double max_x, double min_y, double max_y, double x);
/* the 6x6 information matrix (row-major) of a graph edge from its fitness score: :19-24 (constant) / :30-43 (weights) */
int    mrgfe_inf_matrix_from_fitness(const mrgfe_inf_params* params, double fitness_score, double inf[36]);
/* replaces InformationMatrixCalculator::calc_information_matrix(cloud1, cloud2, relpose) (:14-44; call sites
 * src/mrg_slam/graph_database.cpp:139-142 for every odometry edge, :579-581 for every loop edge): fitness score with the header's
 * default max_range (DBL_MAX), then the weights.  fitness_out (may be NULL) receives the score. */
int    mrgfe_calc_information_matrix(mrgfe_ctx* ctx, const mrgfe_inf_params* params, const float* cloud1, size_t n1, const float* cloud2, size_t n2, size_t stride_bytes,
                                     const double relpose[16], double inf[36], double* fitness_out);

/* ---- per-point passes around the path ------------------------------------------------ */
/* replaces MapCloudGenerator::generate (src/mrg_slam/map_cloud_generator.cpp:14-86) including its
 * pcl::ApproximateMeanVoxelGrid pass (include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126): every keyframe cloud is
 * moved by its pose (poses: n_keyframes column-major 4x4 doubles, Eigen::Isometry3d::matrix()), points farther than
 * distance_far_thresh (if > 0) from their sensor are dropped, the union is voxel-filtered (true mean of x, y, z,
 * intensity per voxel of edge `resolution`, voxels with fewer than min_points_per_voxel points dropped; resolution <= 0
 * returns the unfiltered union).  first_keyframe (may be NULL) marks the clouds skip_first_cloud leaves out.
 * Output order: ascending voxel index (z, y, x) — the reference emits boost::unordered_map order, which is unspecified.
 * Returns MRGFE_ERR_EMPTY where the reference returns nullptr (no keyframes; or nothing left although n_keyframes > 1),
 * MRGFE_ERR_INVALID with *out_n = needed points when capacity is too small. */
int mrgfe_map_cloud_generate(mrgfe_ctx* ctx, int n_keyframes, const float* const* clouds_xyzi, const size_t* n_points, size_t stride_bytes, const double* poses,
                             const uint8_t* first_keyframe, float resolution, int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud,
                             float* out_xyzi, size_t capacity, size_t* out_n);
/* The same over keyframe clouds that stay in HBM.  The map is regenerated from ALL keyframes whenever it is published or
 * saved (apps/mrg_slam_component.cpp:727,781,1097) and only their poses change between calls (graph optimisation), while
 * the reference walks the host clouds every time.  A map store keeps each keyframe's packed cloud resident (append-only,
 * 16 B/point, no cap: the keyframes of a session fit 288 GB many times over); generate() then moves K poses instead of
 * the clouds.  Same output as mrgfe_map_cloud_generate for the same keyframes in the same order. */
typedef struct mrgfe_map_store mrgfe_map_store;
int    mrgfe_map_store_create(mrgfe_ctx* ctx, mrgfe_map_store** out);
void   mrgfe_map_store_destroy(mrgfe_map_store* store);
/* adds keyframe `key` (non-zero); adding a key again with the same point count is a no-op, with another count an error */
int    mrgfe_map_store_add(mrgfe_map_store* store, uint64_t key, const float* xyzi, size_t n, size_t stride_bytes);
int    mrgfe_map_store_has(const mrgfe_map_store* store, uint64_t key, size_t* n);
size_t mrgfe_map_store_bytes(const mrgfe_map_store* store);
int    mrgfe_map_store_generate(mrgfe_map_store* store, int n_keyframes, const uint64_t* keys, const double* poses, const uint8_t* first_keyframe, float resolution,
                                int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud, float* out_xyzi, size_t capacity, size_t* out_n);

/* The same two calls for keyframes that already sit in a map store: a graph edge names its two keyframes, so
 * neither cloud is uploaded again, and the exact-NN grid of key1's cloud is kept for the next edges of that keyframe (the
 * reference builds a fresh kd-tree over cloud1 for every edge: information_matrix_calculator.cpp:51-52). */
int mrgfe_map_store_fitness(mrgfe_map_store* store, uint64_t key1, uint64_t key2, const double relpose[16], double max_range, double* out);
int mrgfe_map_store_information_matrix(mrgfe_map_store* store, const mrgfe_inf_params* params, uint64_t key1, uint64_t key2, const double relpose[16], double inf[36],
                                       double* fitness_out);

/* replaces the other-robot point removal of apps/mrg_slam_component.cpp:396-429: drops every point whose squared float
 * distance to one of the centres (sensor frame, <= 64) is < radius_sqr; kept / removed (may be NULL) keep the input order */
int mrgfe_remove_points_near(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, const float* centres_xyz, int n_centres, float radius_sqr,
                             float* kept_xyzi, size_t* n_kept, float* removed_xyzi, size_t* n_removed);
/* replaces PrefilteringComponent::deskewing (apps/prefiltering_component.cpp:231-292): point i is rotated by the inverse of
 * Quaternionf(1, dt/2 * -w) with dt = scan_period * i / n and w the IMU angular velocity */
int mrgfe_deskew(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, const float ang_v_xyz[3], double scan_period, float* out_xyzi);
/* replaces pcl_ros::transformPointCloud(*src_cloud, *transformed, transform) of PrefilteringComponent::cloud_callback
 * (apps/prefiltering_component.cpp:141: the scan into base_link_frame) = pcl::transformPointCloud with the Matrix4f of the transform:
 * T column-major 4x4 float; non-finite points pass through unchanged, intensity is copied */
int mrgfe_transform_cloud(mrgfe_ctx* ctx, const float* xyzi, size_t n, size_t stride_bytes, const float T[16], float* out_xyzi);

/* ---- batched candidate matching (LoopDetector::matching candidate loop, src/mrg_slam/loop_detector.cpp:126-145) ---- */
typedef struct mrgfe_pair_result {
    float   T[16];      /* final transformation, column-major                */
    double  H[36];      /* 6x6 Hessian of the last iteration, row-major      */
    double  fitness;    /* getFitnessScore(max_range); DBL_MAX if not computed */
    double  trans_probability;
    int32_t converged;
    int32_t iterations;
    int32_t evaluations;
    int32_t pair_id;
} mrgfe_pair_result; /* 384 bytes: the record the ranks all-gather over RCCL */

typedef struct mrgfe_batch mrgfe_batch;
/* A batch holds `n_targets` target clouds and `n_pairs` (target index, source cloud, guess) alignments.  NDT_HIP: they are
 * advanced together, one launch per derivative evaluation for all pairs still running.  GICP_HIP / SMALL_GICP_HIP / VGICP_HIP:
 * the candidates of a target share its covariances and correspondence grid or voxel map (computed once, like the single
 * setInputTarget of loop_detector.cpp:104), the source covariances are computed on a few parallel streams (or taken from the
 * keyframe store below), and the Levenberg-Marquardt loops of all pairs advance together. */
int  mrgfe_batch_create(mrgfe_ctx* ctx, const mrgfe_reg_params* params, mrgfe_batch** out);
void mrgfe_batch_destroy(mrgfe_batch* b);
int  mrgfe_batch_clear(mrgfe_batch* b);
/* returns the target index (>= 0) or an error (< 0) */
int  mrgfe_batch_add_target(mrgfe_batch* b, const float* xyzi, size_t n, size_t stride_bytes);
int  mrgfe_batch_add_target_device(mrgfe_batch* b, const void* d_xyzi, size_t n);
/* returns the pair index (>= 0) or an error (< 0) */
int  mrgfe_batch_add_pair(mrgfe_batch* b, int target_index, const float* src_xyzi, size_t n, size_t stride_bytes, const float guess[16]);
int  mrgfe_batch_add_pair_device(mrgfe_batch* b, int target_index, const void* d_src_xyzi, size_t n, const float guess[16]);
/* n_targets device clouds and n_pairs alignments against them in one call — what a loop over mrgfe_batch_add_target_device /
 * mrgfe_batch_add_pair_device does (pair_target[i] indexes the targets of THIS call; guesses: n_pairs column-major 4x4).
 * Returns the index of the first pair added (the others follow in order) or an error (< 0). */
int  mrgfe_batch_add_device(mrgfe_batch* b, int n_targets, const void* const* d_targets, const size_t* target_points, int n_pairs, const int32_t* pair_target,
                            const void* const* d_sources, const size_t* source_points, const float* guesses);
/* Keyframe store.  The candidates of LoopDetector::matching are old keyframes that come back call after call
 * (loop_detector.cpp:66-100 selects them from the same pool for every new keyframe), while the reference hands their
 * clouds to setInputSource from host memory each time (:128).  A pair added with a non-zero `cloud_key` (the keyframe id)
 * keeps its packed cloud — and, for the GICP methods, its k-NN covariances — resident in HBM inside the batch object,
 * across mrgfe_batch_clear: the next pair with the same key and point count uses them and `src_xyzi` may be NULL.
 * The store is bounded (MRGFE_KEYFRAME_STORE_MB, default 16384; least recently used keyframes not referenced by the
 * current batch are dropped first); mrgfe_batch_forget drops one key (0: all).  Results are identical to the unkeyed call. */
int  mrgfe_batch_add_pair_keyed(mrgfe_batch* b, int target_index, uint64_t cloud_key, const float* src_xyzi, size_t n, size_t stride_bytes,
                                const float guess[16]);
/* 1 if the store holds that key (a keyed add with the same point count may then pass NULL), else 0; *n = its point count */
int  mrgfe_batch_has_cloud(const mrgfe_batch* b, uint64_t cloud_key, size_t* n);
size_t mrgfe_batch_store_bytes(const mrgfe_batch* b);
int  mrgfe_batch_forget(mrgfe_batch* b, uint64_t cloud_key);
int  mrgfe_batch_set_guess(mrgfe_batch* b, int pair_index, const float guess[16]);
/* build every target grid (setInputTarget), then align every pair; fitness_max_range < 0 skips getFitnessScore */
int  mrgfe_batch_build_targets(mrgfe_batch* b);
int  mrgfe_batch_align(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results /* n_pairs */);
/* The same align on a worker thread of the batch's own: returns as soon as the worker holds the batch's context, mrgfe_batch_wait returns the align's
 * status (and sets mrgfe_last_error() in the waiting thread).  `results` must stay valid until then; every other call on this batch waits for the
 * align to finish.  A caller that keeps TWO batches on two contexts in flight — queue batch k + 1 while batch k aligns — fills one batch's target
 * build and straggler rounds with the other's derivative launches: config[1], 256 pairs per batch, 10.2 -> 9.3 ms per batch on one MI355X, same
 * records (a pair's record does not depend on what else runs on the chip).  One align in flight per batch: a second _async before _wait is
 * MRGFE_ERR_STATE. */
int  mrgfe_batch_align_async(mrgfe_batch* b, double fitness_max_range, mrgfe_pair_result* results /* n_pairs */);
int  mrgfe_batch_wait(mrgfe_batch* b);
int  mrgfe_batch_num_pairs(const mrgfe_batch* b);
/* Accounting of the NDT derivative kernel in the last mrgfe_batch_align / mrgfe_reg_align, per kind of evaluation
 * `mode` (0: score+gradient+Hessian, 1: score+gradient, 2: f64 Hessian; -1: all): device time in ms from HIP events recorded
 * around each launch on the context stream, launch count, and the algorithmic bytes of SURVEY.md §8(d): sum over launches and
 * active pairs of N_src*(16 + probes*8) + valid_neighbours*48.  With the default ONE launch per round for all kinds
 * (ndt_derivatives_all_kernel, see mrgfe_dbg_set_fused_launch) time and launch count are reported under mode 0 (and -1), the bytes
 * still per kind; with one launch per kind (ndt_derivatives_kernel<0|1|2,*>) everything is per kind.
 * For GICP_HIP registrations `mode` is ignored and the linearize kernel is reported. */
int  mrgfe_batch_kernel_stats(const mrgfe_batch* b, int mode, double* deriv_ms, int64_t* deriv_launches, double* deriv_alg_bytes);
int  mrgfe_reg_kernel_stats(const mrgfe_reg* reg, int mode, double* deriv_ms, int64_t* deriv_launches, double* deriv_alg_bytes);
/* source points and valid (point, voxel) pairs of the derivative evaluations the last mrgfe_batch_align launched: their ratio
 * is the k-bar of SURVEY.md §8(d) (evaluations answered from a controller's cache are not launched and not counted) */
int  mrgfe_batch_pair_counts(const mrgfe_batch* b, int mode, double* points, double* neighbours);
/* the longest timed derivative launch of the last mrgfe_batch_align: out[0] = device ms (HIP events), out[1..3] = pairs of each evaluation kind
 * (score+gradient+Hessian, score+gradient, f64 Hessian) busy in its round — the launch a rocprofv3 trace shows as the kernel's maximum
 * (diagnostic, like the two calls above: no reference counterpart) */
int  mrgfe_batch_largest_launch(const mrgfe_batch* b, double out[4]);
/* getFitnessScore passes of the last mrgfe_batch_align(fitness_max_range >= 0) of this batch, all launches added up (finished pairs are
 * scored in waves beside the remaining alignment rounds, the rest afterwards): the layout of mrgfe_ctx_fitness_stats, out[10] = launches. */
int  mrgfe_batch_fitness_stats(const mrgfe_batch* b, double out[11]);

/* ---- the same batch over the GPUs of one node ---------------------------------------------------------------------------
 * LoopDetector::matching runs in ONE host process per robot (src/mrg_slam/loop_detector.cpp:104,126-145 under mrg_slam_component's main
 * thread mutex): a node object lets that process reach every GPU of the machine through this header.  It owns one MEMBER per entry of
 * device_ids — a context, a batch and a host thread each; the same ordinal may appear several times (members sharing a card).
 * mrgfe_node_align cuts the declared pair list into n_members contiguous blocks whose sizes differ by at most one (the list is in the
 * reference's order — new keyframe after new keyframe, candidates in candidate order — so a block touches few distinct targets), every
 * member builds the targets and aligns the pairs of its block on its GPU (no data-path collective), and the 384-byte records are gathered
 * into `results` in pair order: with one ncclAllGather over the members' streams (RCCL over xGMI; librccl is loaded with dlopen) when there
 * are two or more members on distinct devices, through host memory otherwise (MRGFE_NODE_GATHER=host / rccl forces either).  The records —
 * and so the best candidates — are those of ONE batch holding the whole list, bit for bit, whatever n_members is.
 * Unlike mrgfe_batch_add_*, the clouds are only REFERENCED by the add calls (which member a pair goes to is known once the list is complete):
 * the host buffers must stay valid until mrgfe_node_align returns.  Keys (non-zero keyframe ids) keep a cloud resident on the member that
 * used it — candidates in the member's batch store (mrgfe_batch_add_pair_keyed), targets in the node's own per-member store —, and a later
 * call that names a resident key with the same point count may pass NULL.  A member that fails (out of memory, a bad cloud) makes
 * mrgfe_node_align return its error code with the member named in mrgfe_last_error(); the node stays usable. */
typedef struct mrgfe_node mrgfe_node;
int    mrgfe_node_create(int n_members, const int* device_ids, const mrgfe_reg_params* params, mrgfe_node** out);
void   mrgfe_node_destroy(mrgfe_node* node);
int    mrgfe_node_num_members(const mrgfe_node* node);
int    mrgfe_node_clear(mrgfe_node* node);
/* registration_->setInputTarget(new_keyframe->cloud), loop_detector.cpp:104: returns the target index (>= 0) or an error (< 0) */
int    mrgfe_node_add_target(mrgfe_node* node, const float* xyzi, size_t n, size_t stride_bytes);
int    mrgfe_node_add_target_keyed(mrgfe_node* node, uint64_t cloud_key, const float* xyzi, size_t n, size_t stride_bytes);
/* one candidate: setInputSource(candidate->cloud) + align(guess), loop_detector.cpp:127-134: returns the pair index (>= 0) or an error (< 0) */
int    mrgfe_node_add_pair(mrgfe_node* node, int target_index, const float* src_xyzi, size_t n, size_t stride_bytes, const float guess[16]);
int    mrgfe_node_add_pair_keyed(mrgfe_node* node, int target_index, uint64_t cloud_key, const float* src_xyzi, size_t n, size_t stride_bytes, const float guess[16]);
int    mrgfe_node_num_pairs(const mrgfe_node* node);
int    mrgfe_node_align(mrgfe_node* node, double fitness_max_range, mrgfe_pair_result* results /* n_pairs, pair_id = index in the list */);
/* block of member `member` in the last mrgfe_node_align; how its records were gathered (0 host memory, 1 RCCL all-gather) */
int    mrgfe_node_shard(const mrgfe_node* node, int member, int* first_pair, int* n_pairs);
int    mrgfe_node_last_gather(const mrgfe_node* node);
int    mrgfe_node_forget(mrgfe_node* node, uint64_t cloud_key); /* drops a key from every member's stores (0: all) */
size_t mrgfe_node_store_bytes(const mrgfe_node* node);
/* the reference's sequential rule on gathered records (loop_detector.cpp:126-145: "if( !hasConverged() || score > best_score ) continue;"): group g
 * = records group_first[g] .. group_first[g + 1] - 1 (the candidates of one new keyframe, in candidate order); best[g] = position within the
 * group or -1, best_score[g] = its fitness or DBL_MAX.  Among equal scores the LAST candidate wins, as there. */
int    mrgfe_node_select_best(const mrgfe_pair_result* results, int n_groups, const int32_t* group_first, int32_t* best, double* best_score);
/* rounds (plan -> derivative launches -> reduce / controller step) of the last mrgfe_batch_align of an NDT_HIP batch */
int mrgfe_batch_rounds(const mrgfe_batch* b);
/* The reference's own performance counters (loop_detector.cpp:22-34: per new keyframe with candidates the wall microseconds of find_candidates + matching
 * and the candidate count; apps/mrg_slam_component.cpp:1032-1037 writes their ratio as average_time_per_candidate_us) for this batch object:
 * out[0] = average_time_per_candidate_us = wall time of all mrgfe_batch_align calls so far (queueing to records, microseconds) / pairs aligned so far,
 * out[1] = wall microseconds of the LAST align, out[2] = its pairs, out[3] = pairs aligned so far.  mrgfe_batch_timing_reset zeroes the totals. */
int mrgfe_batch_timing(const mrgfe_batch* b, double out[4]);
int mrgfe_batch_timing_reset(mrgfe_batch* b);

/* Diagnostic entry points (alternative paths of the same arithmetic, primitives, the optimiser stepped by hand) are declared in mrgfe_debug.h; the two
 * fault injectors there exist only in a library built with -DMRGFE_TESTING (mrg_slam_amd/libmrgfe_testing.so): the shipped libmrgfe.so has neither. */

#ifdef __cplusplus
}
#endif
#endif /* MRGFE_H */

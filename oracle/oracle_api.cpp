// oracle/oracle_api.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the shipped product path.
//
// extern "C" surface of the CPU oracle, bound with ctypes by oracle/oracle.py. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
// Matrices cross this boundary column-major (Eigen default), like the product C-ABI in include/mrgfe.h.
#include <cstring>
#include <vector>

#include "filters.h"
#include "gicp.h"
#include "pcl_gicp.h"
#include "linalg.h"
#include "mapcloud.h"
#include "ndt.h"
#include "pcl_ndt.h"
#include "nn.h"

using namespace orc;

extern "C" {

// ---- filters -----------------------------------------------------------------------------------------
int orc_distance_filter(const float* in, int n, double near_thresh, double far_thresh, float* out) { return distance_filter(in, n, near_thresh, far_thresh, out); }
int orc_approx_voxelgrid(const float* in, int n, float leaf, float* out) { return approx_voxelgrid(in, n, leaf, out); }
int orc_voxelgrid(const float* in, int n, float leaf, int min_pts, int order_mode, float* out, int* out_n) { return voxelgrid(in, n, leaf, min_pts, order_mode, out, out_n); }
int orc_radius_outlier(const float* in, int n, double radius, int min_neighbors, float* out, unsigned char* keep) { return radius_outlier(in, n, radius, min_neighbors, out, keep); }
int orc_statistical_outlier(const float* in, int n, int mean_k, double stddev_mul, float* out, unsigned char* keep) { return statistical_outlier(in, n, mean_k, stddev_mul, out, keep); }

// ---- map cloud, other-robot removal, deskewing ------------------------------------------------
int orc_map_cloud_generate(int K, const float* const* clouds, const int* n, const double* poses, const unsigned char* first_keyframe, float resolution,
                           int min_points_per_voxel, float distance_far_thresh, int skip_first_cloud, float* out, int* out_n)
{
    return map_cloud_generate(K, clouds, n, poses, first_keyframe, resolution, min_points_per_voxel, distance_far_thresh, skip_first_cloud, out, out_n);
}
int orc_remove_points_near(const float* in, int n, const float* centres_xyz, int K, float radius_sqr, float* out, float* removed, int* n_removed)
{
    return remove_points_near(in, n, centres_xyz, K, radius_sqr, out, removed, n_removed);
}
void orc_deskew(const float* in, int n, const float ang_v_xyz[3], double scan_period, float* out) { deskew(in, n, ang_v_xyz, scan_period, out); }

// ---- nearest neighbour ---------------------------------------------------------------------------------
// exact k-NN of each query in the target cloud (grid) and brute force (O(n*m)) for cross-checking the grid
void orc_knn(const float* target, int nt, const float* query, int nq, int k, int* idx, float* sqd)
{
    NnGrid g;
    g.build(target, nt, 1.0f);
    for (int i = 0; i < nq; ++i) {
        int got = g.knn(query[4 * i], query[4 * i + 1], query[4 * i + 2], k, idx + static_cast<size_t>(i) * k, sqd + static_cast<size_t>(i) * k);
        for (int j = got; j < k; ++j) { idx[static_cast<size_t>(i) * k + j] = -1; sqd[static_cast<size_t>(i) * k + j] = -1.0f; }
    }
}
void orc_nn1_brute(const float* target, int nt, const float* query, int nq, int* idx, float* sqd)
{
    for (int i = 0; i < nq; ++i) {
        int best = -1; float bd = 0;
        for (int j = 0; j < nt; ++j) {
            float d = sqdist_f(target[4 * j], target[4 * j + 1], target[4 * j + 2], query[4 * i], query[4 * i + 1], query[4 * i + 2]);
            if (best < 0 || d < bd) { best = j; bd = d; }
        }
        idx[i] = best; sqd[i] = bd;
    }
}
// InformationMatrixCalculator::calc_fitness_score (information_matrix_calculator.cpp:46-81):
// cloud1 = tree, cloud2 transformed by relpose (cast to float), squared NN distance vs un-squared max_range.
double orc_calc_fitness_score(const float* cloud1, int n1, const float* cloud2, int n2, const double relpose_colmajor[16], double max_range)
{
    if (n1 == 0 || n2 == 0) return std::numeric_limits<double>::max();
    float Tc[16], T[16];
    for (int i = 0; i < 16; ++i) Tc[i] = static_cast<float>(relpose_colmajor[i]);
    colmajor_to_rowmajor4(Tc, T);
    NnGrid g;
    g.build(cloud1, n1, 1.0f);
    double sum = 0; int nr = 0;
    for (int i = 0; i < n2; ++i) {
        float x, y, z, d;
        transform_point_f(T, cloud2[4 * i], cloud2[4 * i + 1], cloud2[4 * i + 2], x, y, z);
        if (g.nearest(x, y, z, d) < 0) continue;
        if (static_cast<double>(d) <= max_range) { sum += d; ++nr; }
    }
    return nr > 0 ? sum / nr : std::numeric_limits<double>::max();
}

// ---- small linear algebra (exposed so tests can pin it against numpy) --------------------------------
// InformationMatrixCalculator::weight (/root/reference/src/mrg_slam/information_matrix_calculator.cpp:83-88)
double orc_inf_weight(double a, double max_x, double min_y, double max_y, double x)
{
    double y = (1.0 - std::exp(-a * x)) / (1.0 - std::exp(-a * max_x));
    return min_y + (max_y - min_y) * y;
}
// InformationMatrixCalculator::calc_information_matrix after its fitness score (:19-43): params = {use_const, const_stddev_x, const_stddev_q,
// var_gain_a, min_stddev_x, max_stddev_x, min_stddev_q, max_stddev_q, fitness_score_thresh}; inf row-major 6x6
void orc_inf_matrix(const double params[9], double fitness_score, double inf[36])
{
    for (int k = 0; k < 36; ++k) inf[k] = 0.0;
    for (int k = 0; k < 6; ++k) inf[k * 7] = 1.0;
    double w_x, w_q;
    if (params[0] != 0.0) {
        w_x = params[1];
        w_q = params[2];
    } else {
        double min_var_x = std::pow(params[4], 2), max_var_x = std::pow(params[5], 2), min_var_q = std::pow(params[6], 2), max_var_q = std::pow(params[7], 2);
        w_x = orc_inf_weight(params[3], params[8], min_var_x, max_var_x, fitness_score);
        w_q = orc_inf_weight(params[3], params[8], min_var_q, max_var_q, fitness_score);
    }
    for (int k = 0; k < 3; ++k) { inf[k * 7] /= w_x; inf[(k + 3) * 7] /= w_q; }
}

void orc_svd6_solve(const double A_rowmajor[36], const double b[6], double x[6], double sing[6])
{
    JacobiSvd6 sv; sv.compute(A_rowmajor); sv.solve(b, x);
    for (int i = 0; i < 6; ++i) sing[i] = sv.rank_ >= 0 ? sv.S[i] : 0.0;
}
void orc_sym_eig3(const double A[9], double evals[3], double evecs[9]) { sym_eig3(A, evals, evecs); }
void orc_euler_xyz(const float T_colmajor[16], float out[3]) { float T[16]; colmajor_to_rowmajor4(T_colmajor, T); euler_xyz_f(T, out); }
void orc_pose_to_matrix(const double p[6], float T_colmajor[16]) { float T[16]; pose_to_matrix_f(p, T); rowmajor_to_colmajor4(T, T_colmajor); }
void orc_transform_points(const float T_colmajor[16], const float* in, int n, float* out)
{
    float T[16]; colmajor_to_rowmajor4(T_colmajor, T);
    for (int i = 0; i < n; ++i) { transform_point_f(T, in[4 * i], in[4 * i + 1], in[4 * i + 2], out[4 * i], out[4 * i + 1], out[4 * i + 2]); out[4 * i + 3] = in[4 * i + 3]; }
}

// ---- NDT -------------------------------------------------------------------------------------------------
void* orc_ndt_create() { return new Ndt(); }
void  orc_ndt_destroy(void* h) { delete static_cast<Ndt*>(h); }
void  orc_ndt_set_params(void* h, double resolution, double step_size, double outlier_ratio, double trans_eps, int max_iterations, int num_threads, int search)
{
    Ndt* n = static_cast<Ndt*>(h);
    n->resolution = static_cast<float>(resolution); n->step_size = step_size; n->outlier_ratio = outlier_ratio; n->trans_eps = trans_eps;
    n->max_iterations = max_iterations; n->num_threads = num_threads > 0 ? num_threads : 1; n->search = static_cast<NdtSearch>(search);
}
void orc_ndt_set_gpu_order(void* h, int ppt) { static_cast<Ndt*>(h)->gpu_order_ppt = ppt; }
void orc_ndt_set_fused(void* h, int fused) { static_cast<Ndt*>(h)->fused = fused != 0; }
void orc_ndt_set_thread_sums(void* h, int on) { static_cast<Ndt*>(h)->thread_sums = on != 0; }
int  orc_ndt_set_target(void* h, const float* xyzi, int n) { return static_cast<Ndt*>(h)->set_target(xyzi, n); }
void orc_ndt_set_source(void* h, const float* xyzi, int n) { static_cast<Ndt*>(h)->set_source(xyzi, n); }
void orc_ndt_align(void* h, const float guess_colmajor[16], float* aligned_or_null)
{
    float g[16]; colmajor_to_rowmajor4(guess_colmajor, g);
    static_cast<Ndt*>(h)->align(g, aligned_or_null);
}
int    orc_ndt_converged(void* h) { return static_cast<Ndt*>(h)->converged ? 1 : 0; }
int    orc_ndt_iterations(void* h) { return static_cast<Ndt*>(h)->nr_iterations; }
int    orc_ndt_evals(void* h) { return static_cast<Ndt*>(h)->n_evals; }
void   orc_ndt_last_pose(void* h, double out[6]) { std::memcpy(out, static_cast<Ndt*>(h)->last_p, sizeof(double) * 6); }
double orc_ndt_mean_neighbours(void* h) { Ndt* n = static_cast<Ndt*>(h); return n->n_evals ? n->neighbours_sum / n->n_evals : 0.0; }
double orc_ndt_trans_probability(void* h) { return static_cast<Ndt*>(h)->trans_probability; }
void   orc_ndt_final(void* h, float out_colmajor[16]) { rowmajor_to_colmajor4(static_cast<Ndt*>(h)->final_, out_colmajor); }
void   orc_ndt_hessian(void* h, double out_rowmajor[36]) { std::memcpy(out_rowmajor, static_cast<Ndt*>(h)->hessian, sizeof(double) * 36); }
double orc_ndt_fitness(void* h, double max_range) { return static_cast<Ndt*>(h)->fitness(max_range); }
double orc_ndt_evaluate(void* h, const float T_colmajor[16], const double p[6], int mode, double grad[6], double hess_rowmajor[36])
{
    float T[16]; colmajor_to_rowmajor4(T_colmajor, T);
    return static_cast<Ndt*>(h)->evaluate(T, p, mode, grad, hess_rowmajor);
}
// ---- pcl::NormalDistributionsTransform (registration_method "NDT", registrations.cpp:115-129) ------------------------------
void* orc_pclndt_create() { return new PclNdt(); }
void  orc_pclndt_destroy(void* h) { delete static_cast<PclNdt*>(h); }
void  orc_pclndt_set_params(void* h, double resolution, double step_size, double outlier_ratio, double trans_eps, int max_iterations, int gpu_order, int num_threads)
{
    PclNdt* n = static_cast<PclNdt*>(h);
    n->resolution = static_cast<float>(resolution); n->step_size = step_size; n->outlier_ratio = outlier_ratio; n->trans_eps = trans_eps;
    n->max_iterations = max_iterations; n->gpu_order = gpu_order; n->num_threads = num_threads > 0 ? num_threads : 1;
}
int  orc_pclndt_set_target(void* h, const float* xyzi, int n) { return static_cast<PclNdt*>(h)->set_target(xyzi, n); }
void orc_pclndt_set_source(void* h, const float* xyzi, int n) { static_cast<PclNdt*>(h)->set_source(xyzi, n); }
void orc_pclndt_align(void* h, const float guess_colmajor[16], float* aligned_or_null)
{
    float g[16]; colmajor_to_rowmajor4(guess_colmajor, g);
    static_cast<PclNdt*>(h)->align(g, aligned_or_null);
}
int    orc_pclndt_converged(void* h) { return static_cast<PclNdt*>(h)->converged ? 1 : 0; }
int    orc_pclndt_iterations(void* h) { return static_cast<PclNdt*>(h)->nr_iterations; }
int    orc_pclndt_evals(void* h) { return static_cast<PclNdt*>(h)->n_evals; }
double orc_pclndt_mean_neighbours(void* h) { PclNdt* n = static_cast<PclNdt*>(h); return n->n_evals ? n->neighbours_sum / n->n_evals : 0.0; }
double orc_pclndt_trans_likelihood(void* h) { return static_cast<PclNdt*>(h)->trans_likelihood; }
void   orc_pclndt_final(void* h, float out_colmajor[16]) { rowmajor_to_colmajor4(static_cast<PclNdt*>(h)->final_, out_colmajor); }
void   orc_pclndt_hessian(void* h, double out_rowmajor[36]) { std::memcpy(out_rowmajor, static_cast<PclNdt*>(h)->hessian, sizeof(double) * 36); }
double orc_pclndt_fitness(void* h, double max_range) { return static_cast<PclNdt*>(h)->fitness(max_range); }
double orc_pclndt_evaluate(void* h, const float T_colmajor[16], const double p[6], int mode, double grad[6], double hess_rowmajor[36])
{
    float T[16]; colmajor_to_rowmajor4(T_colmajor, T);
    return static_cast<PclNdt*>(h)->evaluate(T, p, mode, grad, hess_rowmajor);
}
int  orc_pclndt_num_leaves(void* h) { return static_cast<int>(static_cast<PclNdt*>(h)->cells.leaves.size()); }
void orc_pclndt_leaves(void* h, int* keys, int* nr_points, int* in_search, double* mean3, double* icov9, float* centroid4)
{
    PclNdt* n = static_cast<PclNdt*>(h);
    for (size_t i = 0; i < n->cells.leaves.size(); ++i) {
        const NdtLeaf& L = n->cells.leaves[i];
        keys[i] = L.key; nr_points[i] = L.nr_points; in_search[i] = L.in_search;
        std::memcpy(mean3 + 3 * i, L.mean, 24); std::memcpy(icov9 + 9 * i, L.icov, 72); std::memcpy(centroid4 + 4 * i, L.centroid, 16);
    }
}
// target grid inspection: leaves in ascending key order
int  orc_ndt_num_leaves(void* h) { return static_cast<int>(static_cast<Ndt*>(h)->cells.leaves.size()); }
void orc_ndt_grid(void* h, int min_b[3], int max_b[3], int div_b[3]) { Ndt* n = static_cast<Ndt*>(h); for (int a = 0; a < 3; ++a) { min_b[a] = n->cells.min_b[a]; max_b[a] = n->cells.max_b[a]; div_b[a] = n->cells.div_b[a]; } }
void orc_ndt_leaves(void* h, int* keys, int* nr_points, double* mean3, double* cov9, double* icov9)
{
    Ndt* n = static_cast<Ndt*>(h);
    for (size_t i = 0; i < n->cells.leaves.size(); ++i) {
        const NdtLeaf& L = n->cells.leaves[i];
        keys[i] = L.key; nr_points[i] = L.nr_points;
        std::memcpy(mean3 + 3 * i, L.mean, 24); std::memcpy(cov9 + 9 * i, L.cov, 72); std::memcpy(icov9 + 9 * i, L.icov, 72);
    }
}

// ---- GICP (fast_gicp::FastGICP formulation) --------------------------------------------------------------
void* orc_gicp_create() { return new FastGicp(); }
void  orc_gicp_destroy(void* h) { delete static_cast<FastGicp*>(h); }
void  orc_gicp_set_params(void* h, int k_correspondences, double max_corr_dist, double trans_eps, double rot_eps, int max_iterations, int num_threads)
{
    FastGicp* g = static_cast<FastGicp*>(h);
    g->k_correspondences = k_correspondences; g->max_corr_dist = max_corr_dist; g->trans_eps = trans_eps; g->rot_eps = rot_eps;
    g->max_iterations = max_iterations; g->num_threads = num_threads > 0 ? num_threads : 1;
}
void orc_gicp_set_reciprocal(void* h, int on) { static_cast<FastGicp*>(h)->use_reciprocal = on != 0; }
void orc_gicp_set_double_search(void* h, int on) { static_cast<FastGicp*>(h)->double_search = on != 0; }
void orc_gicp_set_variant(void* h, int variant) { static_cast<FastGicp*>(h)->variant = variant; }  // 1: small_gicp formulation, 2: fast_gicp::FastVGICP
void orc_gicp_set_resolution(void* h, double resolution) { static_cast<FastGicp*>(h)->voxel_resolution = resolution; }
int  orc_gicp_num_voxels(void* h) { FastGicp* g = static_cast<FastGicp*>(h); double H[36], b[6], I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; g->linearize(I, H, b, nullptr); return g->num_voxels(); }
void orc_gicp_set_target(void* h, const float* xyzi, int n) { static_cast<FastGicp*>(h)->set_target(xyzi, n); }
void orc_gicp_set_source(void* h, const float* xyzi, int n) { static_cast<FastGicp*>(h)->set_source(xyzi, n); }
void orc_gicp_align(void* h, const float guess_colmajor[16], float* aligned_or_null)
{
    float g[16]; colmajor_to_rowmajor4(guess_colmajor, g);
    static_cast<FastGicp*>(h)->align(g, aligned_or_null);
}
int    orc_gicp_converged(void* h) { return static_cast<FastGicp*>(h)->converged ? 1 : 0; }
int    orc_gicp_iterations(void* h) { return static_cast<FastGicp*>(h)->nr_iterations; }
void   orc_gicp_final(void* h, float out_colmajor[16]) { rowmajor_to_colmajor4(static_cast<FastGicp*>(h)->final_, out_colmajor); }
void   orc_gicp_hessian(void* h, double out_rowmajor[36]) { std::memcpy(out_rowmajor, static_cast<FastGicp*>(h)->final_hessian, sizeof(double) * 36); }
double orc_gicp_fitness(void* h, double max_range) { return static_cast<FastGicp*>(h)->fitness(max_range); }
void   orc_gicp_covariances(void* h, int which, double* cov9_per_point) { static_cast<FastGicp*>(h)->get_covariances(which, cov9_per_point); }
double orc_gicp_linearize(void* h, const double T_rowmajor[16], double H_rowmajor[36], double b[6], int* n_corr)
{
    return static_cast<FastGicp*>(h)->linearize(T_rowmajor, H_rowmajor, b, n_corr);
}

}  // extern "C"

// ---- pcl::GeneralizedIterativeClosestPoint / pclomp::GICP (pcl_gicp.h) --------------------------------------------------------------
extern "C" {
void* orc_pclgicp_create() { return new orc::PclGicp(); }
void  orc_pclgicp_destroy(void* h) { delete static_cast<orc::PclGicp*>(h); }
void  orc_pclgicp_set_params(void* h, int k, double max_corr_dist, double trans_eps, double rot_eps, int max_iterations, int max_inner_iterations, int whole_gradient_norm, int num_threads)
{
    orc::PclGicp* g = static_cast<orc::PclGicp*>(h);
    g->k_correspondences = k; g->max_corr_dist = max_corr_dist; g->trans_eps = trans_eps; g->rot_eps = rot_eps; g->max_iterations = max_iterations;
    g->max_inner_iterations = max_inner_iterations; g->whole_gradient_norm = whole_gradient_norm; g->num_threads = num_threads;
}
void orc_pclgicp_set_gpu_order(void* h, int on) { static_cast<orc::PclGicp*>(h)->gpu_order = on; }
void orc_pclgicp_set_sum_threads(void* h, int t) { static_cast<orc::PclGicp*>(h)->sum_threads = t > 0 ? t : 1; }
void orc_pclgicp_set_target(void* h, const float* xyzi, int n) { static_cast<orc::PclGicp*>(h)->set_target(xyzi, n); }
void orc_pclgicp_set_source(void* h, const float* xyzi, int n) { static_cast<orc::PclGicp*>(h)->set_source(xyzi, n); }
void orc_pclgicp_align(void* h, const float guess_colmajor[16], float* aligned_or_null)
{
    float g[16];
    orc::colmajor_to_rowmajor4(guess_colmajor, g);
    static_cast<orc::PclGicp*>(h)->align(g, aligned_or_null);
}
int    orc_pclgicp_converged(void* h) { return static_cast<orc::PclGicp*>(h)->converged ? 1 : 0; }
int    orc_pclgicp_iterations(void* h) { return static_cast<orc::PclGicp*>(h)->nr_iterations; }
int    orc_pclgicp_evaluations(void* h) { return static_cast<orc::PclGicp*>(h)->n_evaluations; }
int    orc_pclgicp_inner_steps(void* h) { return static_cast<orc::PclGicp*>(h)->n_inner_steps; }
void   orc_pclgicp_final(void* h, float out_colmajor[16]) { orc::rowmajor_to_colmajor4(static_cast<orc::PclGicp*>(h)->final_, out_colmajor); }
double orc_pclgicp_fitness(void* h, double max_range) { return static_cast<orc::PclGicp*>(h)->fitness(max_range); }
void   orc_pclgicp_covariances(void* h, int which, double* out9) { static_cast<orc::PclGicp*>(h)->get_covariances(which, out9); }
double orc_pclgicp_evaluate(void* h, const float T_colmajor[16], const double x[6], double g[6], int* n_corr)
{
    float T[16];
    orc::colmajor_to_rowmajor4(T_colmajor, T);
    return static_cast<orc::PclGicp*>(h)->evaluate(T, x, g, n_corr);
}
}

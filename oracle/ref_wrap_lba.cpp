// oracle/ref_wrap_lba.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Compiles the reference's local-BA observation loops (and, at the end of the file, its pose-only Gauss-Newton loops)
// TEXTUALLY -- the four `for` loops of
// MapHandler::levMarquardtOptimizationLBA, /root/reference/src/mapHandler.cpp:1358-1431 (points, first pass),
// :1436-1540 (lines, first pass), :1587-1666 and :1668-1772 (iteration pass), cut out of the file where it lies by
// oracle/ref_extract_lba.py into oracle/_ref/*.inc -- inside a harness that supplies the names they use:
//   * Eigen spellings -> oracle/ref_shim/mini_dense.hpp (plain loops; NOT Eigen);
//   * map_points / map_lines / map_keyframes -> records with exactly the members the loops read
//     (point3D, line3D, obs_list, T_kf_w; include/mapFeatures.h:40-99, include/keyFrame.h:49-79);
//   * the stvo-pl helpers the loops call -- cam->projection / getFx / getFy, inverse_se3, robustWeightCauchy,
//     SlamConfig::homogTh -- are NOT in the reference tree ([RECALL]): restated here as in oracle/plslam_oracle.c;
//   * expmap_se3( X.block(6*k,0,6,1) ) (iteration pass, points) -> the pose the caller supplies for local slot k
//     (the harness stores k in the block, so no exponential map is evaluated on this side).
// What this pins: the row algebra (residual, both Jacobians, weight), the index arithmetic and the H / g / err
// accumulation of the reference's own source text, including the iteration pass's quirks (line end points both read
// at stride 3, the literal 1e-7, the un-updated key-frame pose for lines).
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include "mini_dense.hpp"

using namespace std;
typedef mini::Fixed<2, 1> Vector2d;
typedef mini::Fixed<3, 1> Vector3d;
typedef mini::Fixed<6, 1> Vector6d;
typedef mini::Fixed<4, 4> Matrix4d;
typedef mini::Fixed<3, 3> Matrix3d;
typedef mini::Fixed<6, 6> Matrix6d;
typedef mini::MatrixX MatrixXd;
typedef mini::MatrixX VectorXd;
typedef mini::Vector6i Vector6i;

namespace {
struct MapPoint { Vector3d point3D; vector<Vector2d> obs_list; vector<double> sigma_list; };   // sigma_list: read, unused (:1641)
struct MapLine { Vector6d line3D; vector<Vector3d> obs_list; };
struct KeyFrame { Matrix4d T_kf_w; };
// the stvo-pl feature records, with the members the pose-only GN loops read (src/mapHandler.cpp:3331-3426)
struct PointFeature { bool inlier; Vector3d P; Vector2d pl_obs; double sigma2; };
struct LineFeature { bool inlier; Vector3d sP, eP, le_obs; double sigma2; };
struct Camera {
    double fx, fy, cx, cy;
    Vector2d projection(const Vector3d& P) const {      // stvo-pl PinholeStereoCamera::projection [RECALL]
        Vector2d p;
        p(0) = cx + fx * P(0) / P(2);
        p(1) = cy + fy * P(1) / P(2);
        return p;
    }
    double getFx() const { return fx; }
    double getFy() const { return fy; }
};
double g_homog_th = 1e-7;
struct SlamConfig { static double homogTh() { return g_homog_th; } };
vector<Matrix4d> g_slot_pose;                            // poses of the local key-frame slots (iteration pass)

Matrix4d inverse_se3(const Matrix4d& T)                  // stvo-pl [RECALL]: [R^T, -R^T t]
{
    Matrix4d o;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) o(i, j) = T(j, i);
        o(i, 3) = -(T(0, i) * T(0, 3) + T(1, i) * T(1, 3) + T(2, i) * T(2, 3));
    }
    o(3, 0) = o(3, 1) = o(3, 2) = 0.0;
    o(3, 3) = 1.0;
    return o;
}
Matrix4d expmap_se3(const Vector6d& x) { return g_slot_pose.at((size_t)x(0)); }
double robustWeightCauchy(double r) { return 1.0 / (1.0 + r * r); }   // stvo-pl [RECALL]
}  // namespace

extern "C" int ref_lba_accumulate(int iter_pass, const double cam4[4], double homog_th, int Nkf, int Npt, int Nls,
                                  const double* T_map, int n_kf_map, const double* T_slot,
                                  const double* Xw, const double* Lw,
                                  const int32_t* pt_lm, const int32_t* pt_kf_map, const int32_t* pt_kf_loc, const double* pt_uv, int n_pt_obs,
                                  const int32_t* ls_lm, const int32_t* ls_kf_map, const int32_t* ls_kf_loc, const double* ls_l, int n_ls_obs,
                                  double* H_out, double* g_out, double* err_out)
{
    try {
        g_homog_th = homog_th;
        Camera cam_ = {cam4[0], cam4[1], cam4[2], cam4[3]};
        Camera* cam = &cam_;
        vector<KeyFrame*> map_keyframes;
        for (int k = 0; k < n_kf_map; ++k) {
            KeyFrame* kf = new KeyFrame;
            for (int i = 0; i < 16; ++i) kf->T_kf_w.v[i] = T_map[16 * (size_t)k + i];
            map_keyframes.push_back(kf);
        }
        g_slot_pose.assign((size_t)Nkf, Matrix4d());
        for (int k = 0; k < Nkf; ++k)
            for (int i = 0; i < 16; ++i) g_slot_pose[k].v[i] = T_slot[16 * (size_t)k + i];
        vector<MapPoint*> map_points;
        for (int j = 0; j < Npt; ++j) {
            MapPoint* p = new MapPoint;
            for (int i = 0; i < 3; ++i) p->point3D(i) = Xw[3 * (size_t)j + i];
            map_points.push_back(p);
        }
        vector<MapLine*> map_lines;
        for (int j = 0; j < Nls; ++j) {
            MapLine* l = new MapLine;
            for (int i = 0; i < 6; ++i) l->line3D(i) = Lw[6 * (size_t)j + i];
            map_lines.push_back(l);
        }
        vector<Vector6i> pt_obs_list, ls_obs_list;
        for (int o = 0; o < n_pt_obs; ++o) {
            MapPoint* p = map_points.at(pt_lm[o]);
            Vector2d uv;
            uv(0) = pt_uv[2 * (size_t)o];
            uv(1) = pt_uv[2 * (size_t)o + 1];
            Vector6i e = {{pt_lm[o], pt_lm[o], (int)p->obs_list.size(), pt_kf_map[o], pt_kf_loc[o], 1}};
            p->obs_list.push_back(uv);
            p->sigma_list.push_back(1.0);
            pt_obs_list.push_back(e);
        }
        for (int o = 0; o < n_ls_obs; ++o) {
            MapLine* l = map_lines.at(ls_lm[o]);
            Vector3d le;
            for (int i = 0; i < 3; ++i) le(i) = ls_l[3 * (size_t)o + i];
            Vector6i e = {{ls_lm[o], ls_lm[o], (int)l->obs_list.size(), ls_kf_map[o], ls_kf_loc[o], 1}};
            l->obs_list.push_back(le);
            ls_obs_list.push_back(e);
        }
        const int N = 6 * Nkf + 3 * Npt + 6 * Nls;
        VectorXd X = VectorXd::Zero(N), g = VectorXd::Zero(N);
        MatrixXd H = MatrixXd::Zero(N, N);
        for (int k = 0; k < Nkf; ++k) X(6 * k) = (double)k;                   // see expmap_se3 above
        for (int i = 0; i < 3 * Npt; ++i) X(6 * Nkf + i) = Xw[i];
        for (int i = 0; i < 6 * Nls; ++i) X(6 * Nkf + 3 * Npt + i) = Lw[i];
        double err = 0.0;
        if (iter_pass == 0) {
#include "_ref/lba_pt_first.inc"
#include "_ref/lba_ls_first.inc"
        } else if (iter_pass == 1) {
#include "_ref/lba_pt_iter.inc"
#include "_ref/lba_ls_iter.inc"
        } else {
            // iter_pass == 2: the first-pass loops of levMarquardtOptimizationGBA (:2124-2228, :2233-2355), which spell the
            // accumulation through SparseMatrix / SparseVector::coeffRef; the outer H / g are shadowed by such objects
            mini::SparseLike H(N, N), g(N);
            int Npt_obs = 0, Nls_obs = 0;
#include "_ref/gba_pt_first.inc"
#include "_ref/gba_ls_first.inc"
            memcpy(H_out, H.v.data(), sizeof(double) * (size_t)N * N);
            memcpy(g_out, g.v.data(), sizeof(double) * (size_t)N);
            *err_out = err;
            for (size_t k = 0; k < map_keyframes.size(); ++k) delete map_keyframes[k];
            for (size_t k = 0; k < map_points.size(); ++k) delete map_points[k];
            for (size_t k = 0; k < map_lines.size(); ++k) delete map_lines[k];
            return 0;
        }
        memcpy(H_out, H.v.data(), sizeof(double) * (size_t)N * N);
        memcpy(g_out, g.v.data(), sizeof(double) * (size_t)N);
        *err_out = err;
        for (size_t k = 0; k < map_keyframes.size(); ++k) delete map_keyframes[k];
        for (size_t k = 0; k < map_points.size(); ++k) delete map_points[k];
        for (size_t k = 0; k < map_lines.size(); ++k) delete map_lines[k];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// The point and line loops of one pose-only Gauss-Newton iteration, MapHandler::computeRelativePoseGN
// (/root/reference/src/mapHandler.cpp:3331-3368, :3371-3426) or, with robust != 0, the first pair of loops of
// computeRelativePoseRobustGN (:3595-3630, :3633-3689) -- compiled textually like the loops above.  Outputs as the
// reference combines them right behind the loops (:3413-3416): H = H_p + H_l, g = g_p + g_l, e = e_p + e_l.
extern "C" int ref_pose_gn_accumulate(int robust, const double cam4[4], double homog_th, const double* T_inc16,
                                      const double* P, const double* pl_obs, const uint8_t* pt_inlier, int npt,
                                      const double* sPeP, const double* le_obs, const uint8_t* ls_inlier, int nls,
                                      double* H_out, double* g_out, double* e_out, int32_t* n_obs)
{
    try {
        g_homog_th = homog_th;
        Camera cam_ = {cam4[0], cam4[1], cam4[2], cam4[3]};
        Camera* cam = &cam_;
        Matrix4d T_inc;
        for (int i = 0; i < 16; ++i) T_inc.v[i] = T_inc16[i];
        vector<PointFeature*> lc_points;
        vector<LineFeature*> lc_lines;
        for (int i = 0; i < npt; ++i) {
            PointFeature* f = new PointFeature;
            f->inlier = pt_inlier[i] != 0;
            f->sigma2 = 1.0;
            for (int k = 0; k < 3; ++k) f->P(k) = P[3 * (size_t)i + k];
            for (int k = 0; k < 2; ++k) f->pl_obs(k) = pl_obs[2 * (size_t)i + k];
            lc_points.push_back(f);
        }
        for (int i = 0; i < nls; ++i) {
            LineFeature* f = new LineFeature;
            f->inlier = ls_inlier[i] != 0;
            f->sigma2 = 1.0;
            for (int k = 0; k < 3; ++k) {
                f->sP(k) = sPeP[6 * (size_t)i + k];
                f->eP(k) = sPeP[6 * (size_t)i + 3 + k];
                f->le_obs(k) = le_obs[3 * (size_t)i + k];
            }
            lc_lines.push_back(f);
        }
        Matrix6d H_p = Matrix6d::Zero(), H_l = Matrix6d::Zero();
        Vector6d g_p = Vector6d::Zero(), g_l = Vector6d::Zero();
        double e_p = 0.0, e_l = 0.0;
        int N_p = 0, N_l = 0;
        if (!robust) {
#include "_ref/gn_pt.inc"
#include "_ref/gn_ls.inc"
        } else {
#include "_ref/gnr_pt.inc"
#include "_ref/gnr_ls.inc"
        }
        Matrix6d H = H_p + H_l;
        Vector6d g = g_p + g_l;
        memcpy(H_out, H.v.data(), 36 * sizeof(double));
        memcpy(g_out, g.v.data(), 6 * sizeof(double));
        *e_out = e_p + e_l;
        n_obs[0] = N_p;
        n_obs[1] = N_l;
        for (size_t k = 0; k < lc_points.size(); ++k) delete lc_points[k];
        for (size_t k = 0; k < lc_lines.size(); ++k) delete lc_lines[k];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

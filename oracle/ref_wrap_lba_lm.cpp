// oracle/ref_wrap_lba_lm.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Compiles the WHOLE Levenberg-Marquardt body of MapHandler::levMarquardtOptimizationLBA TEXTUALLY --
// /root/reference/src/mapHandler.cpp:1334-1812, everything between the function's opening brace and its write-back section:
// variables and parameters, the first pass, lambda = lambdaLbaLM * Hmax (:1544-1550), the first damped solve and update
// (:1552-1575), the iteration loop with its stop tests and lambda schedule (:1583-1812) -- cut out of the file where it lies by
// oracle/ref_extract_lba.py into oracle/_ref/lba_lm_body.inc, inside a harness that supplies the names it uses:
//   * Eigen spellings -> oracle/ref_shim/mini_dense.hpp (plain loops; NOT Eigen); SparseMatrix<double> = a dense copy,
//     H.sparseView() = the matrix itself, SimplicialLDLT< SparseMatrix<double> > = an envelope LDL^T in REVERSED variable order
//     (the landmark blocks first: no fill outside the arrow) -- a different elimination order than Eigen's AMD-ordered
//     simplicial factorisation, equal to rounding;
//   * map_points / map_lines / map_keyframes -> records with the members the body reads;
//   * stvo-pl's helpers -- cam->projection / getFx / getFy, inverse_se3, expmap_se3, logmap_se3, robustWeightCauchy, the Config /
//     SlamConfig getters -- are NOT in the reference tree ([RECALL]): restated here as in oracle/plslam_oracle.c.
// The body keeps no record of its iterations; the harness reads them where the text constructs its solver: the token
// SimplicialLDLT is a macro that first appends the (lambda, err) in scope -- err already normalised (:1541, :1773) -- to a trace.
// What this pins: the LM CONTROL FLOW of the reference's own text -- first step applied unconditionally, err normalised by
// observations in the first pass and by LANDMARKS in the iterations, lambda growing on success, a rejected step followed by a stop
// at the next iteration's first test, the three stop tests -- for plslam_amd/host/lba_rows.hpp's LbaPlanSolver::optimize.
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

extern "C" {        // oracle/plslam_oracle.c's restatements of stvo-pl's SE(3) maps, compiled into this library too (oracle/Makefile)
void plo_inverse_se3(const double T[16], double Ti[16]);
void plo_expmap_se3(const double x[6], double T[16]);
void plo_logmap_se3(const double T[16], double x[6]);
}

namespace {
struct MapPoint { Vector3d point3D; vector<Vector2d> obs_list; vector<double> sigma_list; };
struct MapLine { Vector6d line3D; vector<Vector3d> obs_list; };
struct KeyFrame { Matrix4d T_kf_w; };
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
struct LmCfg { double homog_th, lambda_lm, lambda_k, min_err_change, min_err; int max_iters; } g_cfg;
struct SlamConfig {
    static double homogTh() { return g_cfg.homog_th; }
    static double lambdaLbaLM() { return g_cfg.lambda_lm; }
    static double lambdaLbaK() { return g_cfg.lambda_k; }
    static int maxItersLba() { return g_cfg.max_iters; }
};
struct Config {
    static double minErrorChange() { return g_cfg.min_err_change; }
    static double minError() { return g_cfg.min_err; }
};
Matrix4d inverse_se3(const Matrix4d& T) { Matrix4d o; plo_inverse_se3(T.v.data(), o.v.data()); return o; }
Matrix4d expmap_se3(const Vector6d& x) { Matrix4d o; plo_expmap_se3(x.v.data(), o.v.data()); return o; }
Vector6d logmap_se3(const Matrix4d& T) { Vector6d o; plo_logmap_se3(T.v.data(), o.v.data()); return o; }
double robustWeightCauchy(double r) { return 1.0 / (1.0 + r * r); }   // stvo-pl [RECALL]

template <class T>
struct SparseMatrix : mini::Dyn {
    SparseMatrix() {}
    SparseMatrix(int r_, int c_) : mini::Dyn(r_, c_) {}
    SparseMatrix& operator=(const mini::Dyn& d) { r = d.r; c = d.c; v = d.v; return *this; }
};
// x = A^-1 b for a symmetric positive definite A: LDL^T of the REVERSED matrix (variable N-1 first) inside its envelope
template <class M>
struct LdltStandIn {
    int n;
    vector<double> L;            // reversed, row-major lower triangle incl. the diagonal D
    vector<int> first;
    explicit LdltStandIn(const M& A) : n(A.r), L((size_t)A.r * A.r, 0.0), first(A.r, 0)
    {
        for (int i = 0; i < n; ++i) {
            int f = i;
            for (int j = 0; j <= i; ++j) {
                const double a = A.v[(size_t)(n - 1 - i) * n + (n - 1 - j)];
                L[(size_t)i * n + j] = a;
                if (a != 0.0 && j < f) f = j;
            }
            first[i] = f;
        }
        for (int j = 0; j < n; ++j) {
            double d = L[(size_t)j * n + j];
            for (int k = first[j]; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k] * L[(size_t)k * n + k];
            L[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; ++i) {
                if (first[i] > j) continue;
                double l = L[(size_t)i * n + j];
                for (int k = max(first[i], first[j]); k < j; ++k) l -= L[(size_t)i * n + k] * L[(size_t)j * n + k] * L[(size_t)k * n + k];
                L[(size_t)i * n + j] = l / d;
            }
        }
    }
    mini::Dyn solve(const mini::Dyn& b) const
    {
        vector<double> x(n);
        for (int i = 0; i < n; ++i) x[i] = b.v[n - 1 - i];
        for (int i = 0; i < n; ++i)
            for (int k = first[i]; k < i; ++k) x[i] -= L[(size_t)i * n + k] * x[k];
        for (int i = 0; i < n; ++i) x[i] /= L[(size_t)i * n + i];
        for (int i = n - 1; i >= 0; --i)
            for (int k = i + 1; k < n; ++k)
                if (first[k] <= i) x[i] -= L[(size_t)k * n + i] * x[k];
        mini::Dyn o(n, 1);
        for (int i = 0; i < n; ++i) o.v[n - 1 - i] = x[i];
        return o;
    }
};
vector<double> g_trace;          // (lambda, err) per solve
}  // namespace

#define SimplicialLDLT g_trace.push_back(lambda); g_trace.push_back(err); LdltStandIn

// One local bundle adjustment as the reference's text runs it.  Key frames: n_kf_map stored poses T_map (the body reads
// map_keyframes[kf_idx_map]->T_kf_w); the Nkf optimised ones enter through X_aux = x_kf (6 Nkf) followed by the landmarks.
// Observation columns as in the Vector6i of :1257-1264.  Outputs: X_out (N), trace (2 doubles per solve: lambda, err; room for
// 2 * max_iters), scal_out = {iters, err, err_prev, lambda, n_solves}.
extern "C" int ref_lba_lm(const double cam4[4], const double cfg6[6], int Nkf, int Npt, int Nls, const double* T_map, int n_kf_map,
                          const double* x_kf, const double* Xw, const double* Lw,
                          const int32_t* pt_lm, const int32_t* pt_kf_map, const int32_t* pt_kf_loc, const double* pt_uv, int n_pt_obs,
                          const int32_t* ls_lm, const int32_t* ls_kf_map, const int32_t* ls_kf_loc, const double* ls_l, int n_ls_obs,
                          double* X_out, double* trace_out, double* scal_out)
{
    try {
        g_cfg.homog_th = cfg6[0]; g_cfg.lambda_lm = cfg6[1]; g_cfg.lambda_k = cfg6[2]; g_cfg.max_iters = (int)cfg6[3];
        g_cfg.min_err_change = cfg6[4]; g_cfg.min_err = cfg6[5];
        g_trace.clear();
        Camera cam_ = {cam4[0], cam4[1], cam4[2], cam4[3]};
        Camera* cam = &cam_;
        vector<KeyFrame*> map_keyframes;
        for (int k = 0; k < n_kf_map; ++k) {
            KeyFrame* kf = new KeyFrame;
            for (int i = 0; i < 16; ++i) kf->T_kf_w.v[i] = T_map[16 * (size_t)k + i];
            map_keyframes.push_back(kf);
        }
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
        vector<int> kf_list((size_t)Nkf, 0), pt_list, ls_list;      // (kf_list: only its size is read by the body)
        vector<double> X_aux;
        for (int i = 0; i < 6 * Nkf; ++i) X_aux.push_back(x_kf[i]);
        for (int i = 0; i < 3 * Npt; ++i) X_aux.push_back(Xw[i]);
        for (int i = 0; i < 6 * Nls; ++i) X_aux.push_back(Lw[i]);
        (void)pt_list; (void)ls_list;
        {
#include "_ref/lba_lm_body.inc"
            for (int i = 0; i < N; ++i) X_out[i] = X(i);
            for (size_t k = 0; k < g_trace.size(); ++k) trace_out[k] = g_trace[k];
            scal_out[0] = (double)iters; scal_out[1] = err; scal_out[2] = err_prev; scal_out[3] = lambda; scal_out[4] = (double)(g_trace.size() / 2);
        }
        for (size_t k = 0; k < map_keyframes.size(); ++k) delete map_keyframes[k];
        for (size_t k = 0; k < map_points.size(); ++k) delete map_points[k];
        for (size_t k = 0; k < map_lines.size(); ++k) delete map_lines[k];
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

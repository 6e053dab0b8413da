// oracle/ref_wrap_map2kf.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Compiles TEXTUALLY, from where the file lies (cut out at build time by oracle/ref_extract_lba.py), the loops of
// the reference's map <-> key-frame matchers that sit on either side of the descriptor match:
//   MapHandler::matchMap2KFPoints   visibility pre-filter /root/reference/src/mapHandler.cpp:545-558,
//                                   geometric gate + bookkeeping :601-629
//   MapHandler::matchMap2KFLines    visibility pre-filter :647-663, gate (SIGNED, no abs) :716-749
// inside a harness with the names they use: Eigen spellings from oracle/ref_shim/mini_dense.hpp, cv::Mat from
// oracle/ref_shim/opencv2/core.hpp (rows are pushed into a descriptor matrix), landmark / feature / key-frame
// records with exactly the members the loops touch, SlamConfig thresholds, cam->projection [RECALL stvo-pl].
// The map mutation of the gate loops (add*Observation, full_graph) is accepted and recorded, so the harness can
// report which landmarks passed.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>
#include "mini_dense.hpp"
#include "opencv2/core.hpp"

using namespace std;
using cv::Mat;
typedef mini::Fixed<2, 1> Vector2d;
typedef mini::Fixed<3, 1> Vector3d;
typedef mini::Fixed<4, 1> Vector4d;
typedef mini::Fixed<6, 1> Vector6d;
typedef mini::Fixed<4, 4> Matrix4d;

namespace {
struct MapPoint {
    int idx; bool local, observed; Vector3d point3D; Mat med_desc; vector<int> kf_obs_list;
    void addMapPointObservation(Mat, int, Vector2d, Vector3d) { observed = true; }
};
struct MapLine {
    int idx; bool local, observed; Vector6d line3D; Mat med_desc; vector<int> kf_obs_list;
    void addMapLineObservation(Mat, int, Vector3d, Vector3d, Vector4d) { observed = true; }
};
struct PointFeature { Vector3d P; Vector2d pl; int idx; };
struct LineFeature { Vector3d sP, eP, le; Vector2d spl, epl; int idx; };
struct KeyFrame { Matrix4d T_kf_w; };
struct Frame { double inv_width, inv_height; };
struct Camera {
    double fx, fy, cx, cy; int width, height;
    Vector2d projection(const Vector3d& P) const { Vector2d p; p(0) = cx + fx * P(0) / P(2); p(1) = cy + fy * P(1) / P(2); return p; }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
};
double g_epip_p, g_epip_l;
struct SlamConfig { static double maxKFEpipP() { return g_epip_p; } static double maxKFEpipL() { return g_epip_l; } };
Mat desc_row() { Mat m; m.create(1, 32, CV_8U); memset(m.ptr(), 0, 32); return m; }
Matrix4d pose(const double* T) { Matrix4d m; for (int i = 0; i < 16; ++i) m.v[i] = T[i]; return m; }
}  // namespace

// kind 0: points (X = n x 3), 1: lines (X = n x 6).  vis[n] = 1 for the landmarks the reference's loop selects; pj = the
// normalised projections it records (n_sel x 2 or n_sel x 4, in selection order).  Returns n_sel.
extern "C" int ref_map_visible(int kind, const double cam6[6], const double* Twf16, const double* X, int n, double inv_w,
                               double inv_h, uint8_t* vis, double* pj)
{
    try {
        Camera cam_ = {cam6[0], cam6[1], cam6[2], cam6[3], (int)cam6[4], (int)cam6[5]};
        Camera* cam = &cam_;
        Frame fr = {inv_w, inv_h};
        Frame* curr_frame = &fr;
        const int kf2_idx = 7;
        Matrix4d Twf = pose(Twf16);
        memset(vis, 0, (size_t)n);
        if (kind == 0) {
            vector<MapPoint*> map_points, map_local_points;
            for (int i = 0; i < n; ++i) {
                MapPoint* p = new MapPoint;
                p->idx = i; p->local = true; p->observed = false; p->med_desc = desc_row(); p->kf_obs_list.push_back(3);
                for (int k = 0; k < 3; ++k) p->point3D(k) = X[3 * (size_t)i + k];
                map_points.push_back(p);
            }
            vector<pair<double, double> > pj_points;
            Mat map_lpt_desc;
#include "_ref/m2kf_pt_vis.inc"
            for (size_t k = 0; k < map_local_points.size(); ++k) {
                vis[map_local_points[k]->idx] = 1;
                pj[2 * k] = pj_points[k].first;
                pj[2 * k + 1] = pj_points[k].second;
            }
            const int ns = (int)map_local_points.size();
            if (map_lpt_desc.rows != ns) return -2;
            for (size_t k = 0; k < map_points.size(); ++k) delete map_points[k];
            return ns;
        }
        vector<MapLine*> map_lines, map_local_lines;
        for (int i = 0; i < n; ++i) {
            MapLine* l = new MapLine;
            l->idx = i; l->local = true; l->observed = false; l->med_desc = desc_row(); l->kf_obs_list.push_back(3);
            for (int k = 0; k < 6; ++k) l->line3D(k) = X[6 * (size_t)i + k];
            map_lines.push_back(l);
        }
        vector<pair<pair<double, double>, pair<double, double> > > pj_lines;
        Mat map_lls_desc;
#include "_ref/m2kf_ls_vis.inc"
        for (size_t k = 0; k < map_local_lines.size(); ++k) {
            vis[map_local_lines[k]->idx] = 1;
            pj[4 * k] = pj_lines[k].first.first;   pj[4 * k + 1] = pj_lines[k].first.second;
            pj[4 * k + 2] = pj_lines[k].second.first; pj[4 * k + 3] = pj_lines[k].second.second;
        }
        const int ns = (int)map_local_lines.size();
        for (size_t k = 0; k < map_lines.size(); ++k) delete map_lines[k];
        return ns;
    } catch (const std::exception&) {
        return -1;
    }
}

// kind 0: points (X = nq x 3 landmarks, obs = nt x 2 key-frame pixels pl), 1: lines (X = nq x 6, obs = nt x 3 line
// equations le).  m12[nq] = matches_12; `matches_in` = the count the matcher returned.  mask[nq] = 1 for the landmarks
// that received the observation; returns the reference's final `matches`.
extern "C" int ref_map2kf_gate(int kind, const double cam6[6], const double* Twf16, const double* X, const int32_t* m12, int nq,
                               const double* obs, int nt, double max_epip, int matches_in, uint8_t* mask)
{
    try {
        Camera cam_ = {cam6[0], cam6[1], cam6[2], cam6[3], (int)cam6[4], (int)cam6[5]};
        Camera* cam = &cam_;
        g_epip_p = g_epip_l = max_epip;
        const int kf2_idx = 7;
        Matrix4d Twf = pose(Twf16);
        KeyFrame kf;
        for (int i = 0; i < 16; ++i) kf.T_kf_w.v[i] = (i % 5 == 0) ? 1.0 : 0.0;
        KeyFrame* curr_kf = &kf;
        vector<vector<int> > full_graph(8, vector<int>(8, 0));
        vector<int> matches_12(m12, m12 + nq);
        int matches = matches_in;
        if (kind == 0) {
            vector<MapPoint*> map_points, map_local_points;
            for (int i = 0; i < nq; ++i) {
                MapPoint* p = new MapPoint;
                p->idx = i; p->local = true; p->observed = false; p->kf_obs_list.push_back(3);
                for (int k = 0; k < 3; ++k) p->point3D(k) = X[3 * (size_t)i + k];
                map_points.push_back(p);
                map_local_points.push_back(p);
            }
            vector<PointFeature*> unmatched_points;
            Mat unmatched_pt_desc;
            for (int j = 0; j < nt; ++j) {
                PointFeature* f = new PointFeature;
                f->idx = -1;
                f->P(0) = 0.1 * j; f->P(1) = -0.2; f->P(2) = 3.0 + j;          // only feeds the viewing direction
                f->pl(0) = obs[2 * (size_t)j]; f->pl(1) = obs[2 * (size_t)j + 1];
                unmatched_points.push_back(f);
                unmatched_pt_desc.push_back(desc_row());
            }
#include "_ref/m2kf_pt_gate.inc"
            for (int i = 0; i < nq; ++i) mask[i] = map_points[i]->observed ? 1 : 0;
            for (size_t k = 0; k < map_points.size(); ++k) delete map_points[k];
            for (size_t k = 0; k < unmatched_points.size(); ++k) delete unmatched_points[k];
            return matches;
        }
        vector<MapLine*> map_lines, map_local_lines;
        for (int i = 0; i < nq; ++i) {
            MapLine* l = new MapLine;
            l->idx = i; l->local = true; l->observed = false; l->kf_obs_list.push_back(3);
            for (int k = 0; k < 6; ++k) l->line3D(k) = X[6 * (size_t)i + k];
            map_lines.push_back(l);
            map_local_lines.push_back(l);
        }
        vector<LineFeature*> unmatched_lines;
        Mat unmatched_ls_desc;
        for (int j = 0; j < nt; ++j) {
            LineFeature* f = new LineFeature;
            f->idx = -1;
            for (int k = 0; k < 3; ++k) { f->sP(k) = 1.0 + k; f->eP(k) = 2.0 + k; f->le(k) = obs[3 * (size_t)j + k]; }
            f->spl(0) = f->spl(1) = f->epl(0) = f->epl(1) = 0.0;
            unmatched_lines.push_back(f);
            unmatched_ls_desc.push_back(desc_row());
        }
#include "_ref/m2kf_ls_gate.inc"
        for (int i = 0; i < nq; ++i) mask[i] = map_lines[i]->observed ? 1 : 0;
        for (size_t k = 0; k < map_lines.size(); ++k) delete map_lines[k];
        for (size_t k = 0; k < unmatched_lines.size(); ++k) delete unmatched_lines[k];
        return matches;
    } catch (const std::exception&) {
        return -1000000;
    }
}

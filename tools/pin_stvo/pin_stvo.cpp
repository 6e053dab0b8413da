// Replays the committed golden vectors of the matcher path through a REAL stvo-pl build (its matching.cpp / gridStructure.cpp /
// config.cpp against OpenCV): StVO::match on every match case, both StVO::matchGrid overloads on every grid case, and reports
// every table entry that differs.  This is what turns "parity unpinned"  into "pinned" on a machine that
// has stvo-pl: the product is not involved, only the expectations the product is tested against.
//
//   python tools/pin_stvo/export_cases.py cases && make -C tools/pin_stvo STVO_PL_DIR=/path/to/stvo-pl && tools/pin_stvo/pin_stvo cases
//
// The stvo-pl interface used (recalled; adjust here if a checkout differs):
//   namespace StVO { int match(const cv::Mat&, const cv::Mat&, float nnr, std::vector<int>&);
//                    int matchGrid(const std::vector<point_2d>&, const cv::Mat&, const GridStructure&, const cv::Mat&,
//                                  const GridWindow&, std::vector<int>&);
//                    int matchGrid(const std::vector<line_2d>&, const cv::Mat&, const GridStructure&, const cv::Mat&,
//                                  const std::vector<std::pair<double, double>>&, const GridWindow&, std::vector<int>&); }
//   Config::bestLRMatches(), Config::minRatio12P(), Config::lineSimTh() return references to the singleton's fields.
// The SE(3) helpers, the pinhole projection and the Cauchy weight (kind=se3; [RECALL] as well) go through se3_adapter.h; they
// are compared to a relative 1e-12 (stvo-pl may order its floating-point operations differently), not bit for bit.
// Output: one PASS / FAIL line per case and, at the end, ONE table with a row per golden file.
// The stereo gates (kind=gate_*) are members of StereoFrame that match and gate in one function: they cannot be called on
// their own.  Their cases are listed (inputs, thresholds, expected tables and disparities are in the export) for a manual
// comparison inside StereoFrame::matchStereoPoints / matchStereoLines.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include <opencv2/core.hpp>

#include "config.h"
#include "gridStructure.h"
#include "matching.h"
#include "se3_adapter.h"

namespace {

struct Array {
    char dtype = 0;
    std::vector<int64_t> dims;
    std::vector<uint8_t> bytes;
    int64_t rows() const { return dims.empty() ? 0 : dims[0]; }
    template <class T> const T* as() const { return reinterpret_cast<const T*>(bytes.data()); }
    template <class T> T* as() { return reinterpret_cast<T*>(bytes.data()); }
};

Array load(const std::string& path)
{
    Array a;
    std::ifstream f(path, std::ios::binary);
    char magic[4];
    int32_t nd = 0;
    if (!f.read(magic, 4) || std::memcmp(magic, "PLSA", 4) != 0) { std::fprintf(stderr, "bad array file %s\n", path.c_str()); std::exit(2); }
    f.read(&a.dtype, 1);
    f.read(reinterpret_cast<char*>(&nd), 4);
    a.dims.resize((size_t)nd);
    f.read(reinterpret_cast<char*>(a.dims.data()), 8 * nd);
    int64_t n = 1;
    for (int64_t d : a.dims) n *= d;
    const int64_t esz = a.dtype == 'u' ? 1 : a.dtype == 'd' ? 8 : 4;
    a.bytes.resize((size_t)(n * esz));
    f.read(reinterpret_cast<char*>(a.bytes.data()), n * esz);
    return a;
}

cv::Mat desc_mat(Array& a) { return cv::Mat((int)a.rows(), 32, CV_8U, a.bytes.data()); }

int compare(const std::string& what, const std::vector<int>& got, const Array& want)
{
    int bad = 0;
    if ((int64_t)got.size() != want.rows()) {
        std::printf("FAIL %s: %zu entries, expected %lld\n", what.c_str(), got.size(), (long long)want.rows());
        return 1;
    }
    for (size_t i = 0; i < got.size(); ++i)
        if (got[i] != want.as<int32_t>()[i]) {
            if (bad < 5) std::printf("  %s: row %zu got %d expected %d\n", what.c_str(), i, got[i], want.as<int32_t>()[i]);
            ++bad;
        }
    std::printf("%s %s%s\n", bad ? "FAIL" : "PASS", what.c_str(), bad ? (" (" + std::to_string(bad) + " rows differ)").c_str() : "");
    return bad ? 1 : 0;
}

// |got - want| <= tol * max(1, |want|) element-wise; non-finite values must agree exactly (inf with inf, NaN with NaN)
int compare_f64(const std::string& what, const double* got, const Array& want, double tol)
{
    int64_t n = 1;
    for (int64_t d : want.dims) n *= d;
    int bad = 0;
    double worst = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double w = want.as<double>()[i], g = got[i];
        bool ok;
        if (std::isnan(w) || std::isnan(g)) ok = std::isnan(w) && std::isnan(g);
        else if (std::isinf(w) || std::isinf(g)) ok = w == g;
        else {
            const double e = std::fabs(g - w) / std::fmax(1.0, std::fabs(w));
            worst = std::fmax(worst, e);
            ok = e <= tol;
        }
        if (!ok && bad++ < 5) std::printf("  %s: element %lld got %.17g expected %.17g\n", what.c_str(), (long long)i, g, w);
    }
    std::printf("%s %s (max relative difference %.3g)%s\n", bad ? "FAIL" : "PASS", what.c_str(), worst,
                bad ? (" (" + std::to_string(bad) + " elements differ)").c_str() : "");
    return bad ? 1 : 0;
}

struct Tally { int ran = 0, failed = 0, listed = 0; };

}  // namespace

int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : "pin_cases";
    std::ifstream mf(dir + "/manifest.txt");
    if (!mf) { std::fprintf(stderr, "no manifest in %s (run tools/pin_stvo/export_cases.py first)\n", dir.c_str()); return 2; }
    int failed = 0, ran = 0, listed = 0;
    std::map<std::string, Tally> table;             // one row per golden file
    std::string line;
    while (std::getline(mf, line)) {
        std::map<std::string, std::string> kv;
        std::istringstream ss(line);
        std::string tok;
        while (ss >> tok) {
            const size_t eq = tok.find('=');
            if (eq != std::string::npos) kv[tok.substr(0, eq)] = tok.substr(eq + 1);
        }
        if (kv.empty()) continue;
        const std::string kind = kv["kind"];
        auto arr = [&](const char* key) { return load(dir + "/" + kv[key]); };
        Tally& row = table[kind == "match" ? "match_golden.npz" : kind.rfind("grid_", 0) == 0 ? "grid_golden.npz"
                           : kind == "se3" ? "se3_helpers_golden.npz" : "stereo_gates_golden.npz"];
        const int failed_before = failed;
        if (kind == "match") {
            Array q = arr("q"), t = arr("t"), want = arr("expect");
            StVO::Config::bestLRMatches() = kv["mutual"] == "1";
            std::vector<int> m12;
            StVO::match(desc_mat(q), desc_mat(t), (float)std::atof(kv["nnr"].c_str()), m12);
            failed += compare("match " + kv["name"] + " nnr " + kv["nnr"] + " mutual " + kv["mutual"], m12, want);
            ++ran; ++row.ran;
        } else if (kind == "grid_points" || kind == "grid_lines") {
            Array cen = arr("centres"), d1 = arr("d1"), d2 = arr("d2"), cs = arr("cell_start"), it = arr("cell_items"),
                  want = arr("expect");
            const int cols = std::atoi(kv["cols"].c_str()), rows = std::atoi(kv["rows"].c_str());
            StVO::GridStructure grid(rows, cols);
            for (int x = 0; x < cols; ++x)                      // cell id = x * rows + y, items in push_back order
                for (int y = 0; y < rows; ++y)
                    for (int32_t k = cs.as<int32_t>()[x * rows + y]; k < cs.as<int32_t>()[x * rows + y + 1]; ++k)
                        grid.at(x, y).push_back(it.as<int32_t>()[k]);
            StVO::GridWindow w;
            w.width = std::make_pair(std::atoi(kv["w0"].c_str()), std::atoi(kv["w1"].c_str()));
            w.height = std::make_pair(std::atoi(kv["w2"].c_str()), std::atoi(kv["w3"].c_str()));
            StVO::Config::bestLRMatches() = kv["mutual"] == "1";
            StVO::Config::minRatio12P() = std::atof(kv["nnr"].c_str());
            std::vector<int> m12;
            const int32_t* c = cen.as<int32_t>();
            if (kind == "grid_points") {
                std::vector<StVO::point_2d> pts;
                for (int64_t i = 0; i < cen.rows(); ++i) pts.push_back(std::make_pair(c[2 * i], c[2 * i + 1]));
                StVO::matchGrid(pts, desc_mat(d1), grid, desc_mat(d2), w, m12);
            } else {
                Array dir2 = arr("dir2");
                StVO::Config::lineSimTh() = 0.75;              // the fixtures' sim_th (tests/test_match_grid_cpu.py line_case)
                std::vector<StVO::line_2d> lns;
                for (int64_t i = 0; i < cen.rows(); ++i)
                    lns.push_back(std::make_pair(std::make_pair(c[4 * i], c[4 * i + 1]), std::make_pair(c[4 * i + 2], c[4 * i + 3])));
                std::vector<std::pair<double, double>> dirs;
                for (int64_t i = 0; i < dir2.rows(); ++i) dirs.push_back(std::make_pair(dir2.as<double>()[2 * i], dir2.as<double>()[2 * i + 1]));
                StVO::matchGrid(lns, desc_mat(d1), grid, desc_mat(d2), dirs, w, m12);
            }
            failed += compare(kind + " " + kv["name"] + " nnr " + kv["nnr"] + " mutual " + kv["mutual"], m12, want);
            ++ran; ++row.ran;
        } else if (kind == "se3") {
            const double tol = 1e-12;
            Array tw = arr("twists"), ex = arr("expmap"), inv = arr("inverse"), lg = arr("logmap"), cam = arr("cam"),
                  pts = arr("points"), pj = arr("projection"), cr = arr("cauchy_r"), cw = arr("cauchy_w");
            const int64_t n = tw.rows();
            std::vector<double> o_ex(16 * n), o_inv(16 * n), o_lg(6 * n), o_pj(2 * pts.rows()), o_cw(cr.rows());
            for (int64_t i = 0; i < n; ++i) {
                pin::expmap_se3(tw.as<double>() + 6 * i, o_ex.data() + 16 * i);
                pin::inverse_se3(ex.as<double>() + 16 * i, o_inv.data() + 16 * i);     // of the EXPECTED pose: the cases stay independent
                pin::logmap_se3(ex.as<double>() + 16 * i, o_lg.data() + 6 * i);
            }
            for (int64_t i = 0; i < pts.rows(); ++i) pin::projection(cam.as<double>(), pts.as<double>() + 3 * i, o_pj.data() + 2 * i);
            for (int64_t i = 0; i < cr.rows(); ++i) o_cw[i] = pin::cauchy(cr.as<double>()[i]);
            failed += compare_f64("se3 expmap_se3", o_ex.data(), ex, tol);
            failed += compare_f64("se3 inverse_se3", o_inv.data(), inv, tol);
            failed += compare_f64("se3 logmap_se3", o_lg.data(), lg, 1e-9);          // (acos near +-1: conditioning, not convention)
            failed += compare_f64("se3 projection", o_pj.data(), pj, tol);
            failed += compare_f64("se3 robustWeightCauchy", o_cw.data(), cw, tol);
            ran += 5; row.ran += 5;
        } else {
            ++listed; ++row.listed;                             // gate_points / gate_lines: see the header comment
        }
        row.failed += failed - failed_before;
    }
    std::printf("\n%-28s %8s %6s %6s %s\n", "golden file", "replayed", "PASS", "FAIL", "listed only");
    for (const auto& kvp : table)
        std::printf("%-28s %8d %6d %6d %d\n", kvp.first.c_str(), kvp.second.ran, kvp.second.ran - kvp.second.failed, kvp.second.failed,
                    kvp.second.listed);
    std::printf("%d cases replayed, %d differ; %d stereo-gate cases listed for manual comparison\n", ran, failed, listed);
    return failed ? 1 : 0;
}

// oracle/ref_wrap_mapfeatures.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Runs the reference's OWN representative-descriptor selection,
//   PLSLAM::MapPoint::updateAverageDescDir   /root/reference/src/mapFeatures.cpp:51-93
//   PLSLAM::MapLine::updateAverageDescDir    /root/reference/src/mapFeatures.cpp:121-163
// as the map does: construct the landmark from its first observation (:30-41, :97-109), add the others one by one
// (:43-49, :111-119: every addition re-selects med_desc).  The Makefile compiles mapFeatures.cpp itself from where it
// lies against the cv:: / Eigen stand-ins under oracle/ref_shim/.  Returns the index of the observation whose
// descriptor ended up as med_desc (the stand-in Mat shares its buffer on copy, so identity = same data pointer).
#include <stdint.h>
#include <string.h>
#include "mapFeatures.h"

static cv::Mat row_of(const uint8_t* d)
{
    cv::Mat m;
    m.create(1, 32, CV_8U);
    memcpy(m.ptr(), d, 32);
    return m;
}

template <class LM>
static int selected(const LM& lm)
{
    for (size_t i = 0; i < lm.desc_list.size(); ++i)
        if (lm.desc_list[i].data == lm.med_desc.data) return (int)i;
    return -1;
}

extern "C" int ref_median_desc_point(const uint8_t* desc, int n)
{
    if (n <= 0) return -1;
    Vector3d p3, dir;
    Vector2d obs;
    for (int i = 0; i < 3; ++i) p3(i) = dir(i) = 0.0;
    obs(0) = obs(1) = 0.0;
    PLSLAM::MapPoint mp(0, p3, row_of(desc), 0, obs, dir);
    for (int i = 1; i < n; ++i) mp.addMapPointObservation(row_of(desc + 32 * (size_t)i), i, obs, dir);
    return selected(mp);
}

extern "C" int ref_median_desc_line(const uint8_t* desc, int n)
{
    if (n <= 0) return -1;
    Vector6d l3;
    Vector3d obs, dir;
    Vector4d pts;
    for (int i = 0; i < 6; ++i) l3(i) = 0.0;
    for (int i = 0; i < 3; ++i) obs(i) = dir(i) = 0.0;
    for (int i = 0; i < 4; ++i) pts(i) = 0.0;
    PLSLAM::MapLine ml(0, l3, row_of(desc), 0, obs, dir, pts);
    for (int i = 1; i < n; ++i) ml.addMapLineObservation(row_of(desc + 32 * (size_t)i), i, obs, dir, pts);
    return selected(ml);
}

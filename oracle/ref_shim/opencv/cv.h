// oracle/ref_shim/opencv/cv.h -- TEST INFRASTRUCTURE ONLY: what /root/reference/include/mapFeatures.h asks of
// <opencv/cv.h>.  Everything lives in the cv:: stand-in opencv2/core.hpp (NOT OpenCV; see its header).
#include "../opencv2/core.hpp"

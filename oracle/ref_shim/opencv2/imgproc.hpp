// oracle/ref_shim -- TEST INFRASTRUCTURE ONLY: empty stand-in, everything lives in opencv2/core.hpp
#include "opencv2/core.hpp"

// compile-only probe: resource usage of the new line-product kernel
#include "../../ripp_amd/csrc/kernels.hpp"
#include "../../ripp_amd/csrc/line_products.hpp"

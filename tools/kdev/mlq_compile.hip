// compile-only probe of the carry-free Miller-line kernel: register / scratch figures without building the whole engine
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null tools/kdev/mlq_compile.hip
#include "../../ripp_amd/csrc/fq_miller.hpp"

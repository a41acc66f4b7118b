// twilight_amd/csrc/host/align_gpu.hpp -- shared by the two level kernels (align_gpu.cpp: host-staged, align_resident.cpp: device-resident).
#pragma once
#include "twl_host.hpp"

#include "../../../include/twl_align.h"

#include <chrono>

namespace msa {
namespace progressive {
namespace gpu {

void ensureInit(Option *option);
const std::vector<int> &selectedDevices();
twl_params baseParams(Params &param);          // == Talco_xdrop::Params(msa::Params&), TALCO-XDrop.cpp:36-53
inline double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

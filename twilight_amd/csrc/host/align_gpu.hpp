// twilight_amd/csrc/host/align_gpu.hpp -- shared by the two level kernels (align_gpu.cpp: host-staged, align_resident.cpp: device-resident).
#pragma once
#include "twl_host.hpp"

#include "../../../include/twl_align.h"
#include "../../../include/twl_level.h"

#include <chrono>

namespace msa {
namespace progressive {
namespace gpu {

void ensureInit(Option *option);

struct RunCtx {
    std::vector<twl_store *> stores;      // device-resident mode: one replica per device (or per virtual device in tests)
    std::vector<int> storeDev;
    bool finished = false;                         // the main pass is over: rows are back on the host
    int nextCacheId = 0;
    LevelTotals totals;
    std::vector<LevelRecord> levels;
    Shard shard;
    struct Raw { char *p = nullptr; size_t cap = 0; bool pinned = false; char *get(size_t n); ~Raw(); Raw() = default; Raw(Raw &&o) noexcept : p(o.p), cap(o.cap), pinned(o.pinned) { o.p = nullptr; o.cap = 0; } Raw(const Raw &) = delete; };      // grow-only, never zero-filled host buffer
    ~RunCtx();
};
// Longest-processing-time deal of a level's pairs to `parts` owners (deterministic: every rank computes the same answer).
std::vector<int> dealPairs(const std::vector<long long> &cost, const std::vector<char> &takesPart, int parts);
// All-gather of the paths the ranks aligned: on return paths/errs hold every pair of the level on every rank.
void exchangePaths(RunCtx &ctx, const std::vector<int> &owner, const std::vector<char> &takesPart, int pathCap, std::vector<alnPath> &paths,
                   std::vector<int16_t> &errs, LevelRecord &rec);
// The same for the device-resident level kernel when the processes can all-gather DEVICE blocks (Shard::exchangeDev): every rank's final
// paths (gappy columns already restored, in the store's path buffer / DP output) are packed into one device block, all-gathered, and the
// other ranks' paths unpacked into their rows of the path buffer -- HBM to HBM, one collective per level.  bound[i] = refLen + qryLen of
// pair i before removal (every rank knows it: the block size is agreed on without a collective).  On return fromDp / dpLen / errs describe
// every pair of the level on every rank (fromDp 2 = row in the path buffer).
void exchangeFinalPaths(RunCtx &ctx, twl_store *store, int device, const twl_params &tp, const std::vector<int> &owner, const std::vector<char> &takesPart,
                        const std::vector<int32_t> &bound, int pathStride, std::vector<uint8_t> &fromDp, std::vector<int32_t> &dpLen, std::vector<int16_t> &errs,
                        LevelRecord &rec);
const std::vector<int> &selectedDevices();
twl_params baseParams(Params &param);          // == Talco_xdrop::Params(msa::Params&), TALCO-XDrop.cpp:36-53
inline double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

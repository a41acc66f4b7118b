// twilight_amd/csrc/host/align_gpu.cpp -- msa::progressive::gpu::alignmentKernel_GPU: the level kernel that drops in
// for cpu::alignmentKernel_CPU (/root/reference/src/alignment-cpu.cpp:32-183).  Same phases, same policy; the per-pair
// Talco_xdrop::Align_freq calls of a level (alignment-cpu.cpp:95-130) become twl_align_batch calls (include/twl_align.h).
// There is no CPU alignment path in this file: if the GPU library fails the run stops.
#include "twl_host.hpp"

#include "../../../include/twl_align.h"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <tuple>

namespace msa {
namespace progressive {
namespace gpu {

LevelTotals g_totals;
static std::vector<int> g_devices;

static void ensureInit(Option *option)
{
    static bool done = false;
    if (done) return;
    int rc = option->gpuIdx.empty() ? twl_init(nullptr, 0) : twl_init(option->gpuIdx.data(), (int)option->gpuIdx.size());
    if (rc != TWL_OK) { std::cerr << "ERROR: twl_init failed: " << twl_last_error() << '\n'; exit(1); }
    g_devices = option->gpuIdx.empty() ? std::vector<int>{0} : option->gpuIdx;
    done = true;
}

static twl_params baseParams(Params &param)          // == Talco_xdrop::Params(msa::Params&), TALCO-XDrop.cpp:36-53
{
    twl_params tp{};
    tp.P = param.matrixSize + 1;
    for (int l = 0; l < param.matrixSize; ++l)
        for (int m = 0; m < param.matrixSize; ++m) tp.matrix[l * param.matrixSize + m] = param.scoringMatrix[l][m];
    tp.gap_open = param.gapOpen;
    tp.gap_extend = param.gapExtend;
    tp.gap_boundary = param.gapBoundary;
    tp.gap_char = param.gapExtend;
    tp.xdrop = static_cast<int32_t>(1000 * -1 * param.gapExtend);
    tp.flen = 1 << 12;
    tp.marker = 1 << 10;
    return tp;
}

// Align the pairs `ids` (indices into `in`) with one parameter set; results land in paths/errs.
static void runBatch(const twl_params &tp, const std::vector<int> &ids, std::vector<PairInputs> &in, int P, std::vector<alnPath> &paths,
                     std::vector<int16_t> &errs)
{
    const int n = (int)ids.size();
    if (n == 0) return;
    int seqLen = 1;
    for (int id : ids) seqLen = std::max({seqLen, in[id].lens.first, in[id].lens.second});
    std::vector<float> freq((size_t)n * 2 * seqLen * P, 0.0f), gop((size_t)n * 2 * seqLen, 0.0f), gex((size_t)n * 2 * seqLen, 0.0f);
    std::vector<int32_t> len(2 * (size_t)n), num(2 * (size_t)n), alnLen(n);
    std::vector<int16_t> err(n);
    std::vector<int8_t> aln((size_t)n * 2 * seqLen);
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < n; ++t) {
        const PairInputs &pi = in[ids[t]];
        for (int side = 0; side < 2; ++side) {
            const int L = side ? pi.lens.second : pi.lens.first;
            const float *src = pi.freq.data() + (size_t)side * P * pi.memLen;
            float *dst = &freq[((size_t)t * 2 + side) * seqLen * P];
            std::copy(src, src + (size_t)L * P, dst);
            std::copy(pi.gapOp.data() + (size_t)side * pi.memLen, pi.gapOp.data() + (size_t)side * pi.memLen + L, &gop[((size_t)t * 2 + side) * seqLen]);
            std::copy(pi.gapEx.data() + (size_t)side * pi.memLen, pi.gapEx.data() + (size_t)side * pi.memLen + L, &gex[((size_t)t * 2 + side) * seqLen]);
        }
        len[2 * t] = pi.lens.first; len[2 * t + 1] = pi.lens.second;
        num[2 * t] = pi.refNum; num[2 * t + 1] = pi.qryNum;
    }
    int rc = twl_align_batch(&tp, n, seqLen, freq.data(), gop.data(), gex.data(), len.data(), num.data(), aln.data(), alnLen.data(), err.data());
    if (rc != TWL_OK) { std::cerr << "ERROR: twl_align_batch failed (" << rc << "): " << twl_last_error() << '\n'; exit(1); }
    for (int dev : g_devices) {
        twl_stats st{};
        if (twl_get_stats(dev, &st) == TWL_OK) { g_totals.band_cells += st.band_cells; g_totals.kernel_ms += st.kernel_ms; g_totals.total_ms += st.total_ms; }
    }
    g_totals.pairs += n;
    for (int t = 0; t < n; ++t) {
        errs[ids[t]] = err[t];
        paths[ids[t]].assign(&aln[(size_t)t * 2 * seqLen], &aln[(size_t)t * 2 * seqLen] + (err[t] == 0 ? alnLen[t] : 0));
    }
}

void alignmentKernel_GPU(Tree *, NodePairVec &nodes, SequenceDB *database, Option *option, Params &param)
{
    if (option->cpuOnly) { std::cerr << "ERROR: --cpu-only is not available: this build has no CPU alignment path.\n"; exit(1); }
    ensureInit(option);
    const int n = (int)nodes.size();
    const int P = param.matrixSize + 1;
    std::vector<PairInputs> in(n);
    // wide levels: one pair per thread; narrow levels (upper tree): pairs in turn, the helpers' own column/sequence loops fan out
    const bool acrossPairs = n >= omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) if (acrossPairs)
    for (int i = 0; i < n; ++i) preparePair(nodes[i], database, option, param, in[i]);       // alignment-cpu.cpp:50-93

    std::vector<alnPath> paths(n);
    std::vector<int16_t> errs(n, 0);
    // pairs that go to the DP, grouped by gapCharScore (alignment-cpu.cpp:88)
    std::vector<int> plain, zeroGap;
    for (int i = 0; i < n; ++i) {
        if (in[i].refLen == 0) paths[i].assign(in[i].qryLen, 1);                              // :89-90
        if (in[i].qryLen == 0) paths[i].insert(paths[i].end(), in[i].refLen, 2);
        if (!paths[i].empty() || in[i].lowQ_r || in[i].lowQ_q) continue;                       // :93,95
        const bool zg = (database->currentTask == 1 || database->currentTask == 2 || in[i].refNum > 10000 || in[i].qryNum > 10000);
        (zg ? zeroGap : plain).push_back(i);
    }
    twl_params tp = baseParams(param);
    runBatch(tp, plain, in, P, paths, errs);
    twl_params tz = tp;
    tz.gap_char = 0;
    runBatch(tz, zeroGap, in, P, paths, errs);

    // alignment-cpu.cpp:108-129: task 0 defers a failed pair; later tasks retry with a larger X-drop / band limit
    std::vector<int> fallbackPairs;
    for (int i = 0; i < n; ++i) {
        if (errs[i] == 0) continue;
        if (errs[i] == 3) { std::cout << "There might be some bugs in the code!\n"; exit(1); }
        if (database->currentTask == 0) { paths[i].clear(); fallbackPairs.push_back(i); continue; }
        twl_params tr = (database->currentTask == 1 || database->currentTask == 2 || in[i].refNum > 10000 || in[i].qryNum > 10000) ? tz : tp;
        const int minLen = std::min(in[i].lens.first, in[i].lens.second);
        while (errs[i] != 0) {
            if (errs[i] == 3) { std::cout << "There might be some bugs in the code!\n"; exit(1); }
            if (errs[i] == 2) tr.flen = std::min(static_cast<int32_t>(tr.flen * 1.2) << 1, minLen);
            else { tr.xdrop = static_cast<int32_t>(tr.xdrop * 2); tr.flen = std::min(static_cast<int32_t>(tr.xdrop * 4) << 1, minLen); }
            if (option->printDetail) std::cout << "Retry pair No. " << i << "\txdrop " << tr.xdrop << " flen " << tr.flen << '\n';
            if (tr.flen > 4096) { std::cerr << "ERROR: retry needs an anti-diagonal limit of " << tr.flen << " > 4096, which this build's kernels do not cover.\n"; exit(1); }
            runBatch(tr, std::vector<int>{i}, in, P, paths, errs);
        }
    }

    std::vector<char> deferred(n, 0);
#pragma omp parallel for schedule(dynamic, 1) if (acrossPairs)
    for (int i = 0; i < n; ++i) {                                                              // alignment-cpu.cpp:136-175
        // low-quality singleton rule (:136-144): such a pair is deferred whatever the DP said
        deferred[i] = (database->currentTask == 0 && (in[i].refNum == 1 || in[i].qryNum == 1) && (in[i].lowQ_r || in[i].lowQ_q)) ? 1 : 0;
        finishPair(nodes[i], database, option, param, in[i], paths[i]);
        in[i] = PairInputs();                                                                   // release the profile buffers early
    }
    for (int i = 0; i < n; ++i)
        if (deferred[i]) fallbackPairs.push_back(i);
    if (!fallbackPairs.empty()) alignment_helper::fallback2cpu(fallbackPairs, nodes, database, option);
}

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

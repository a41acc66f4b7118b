// twilight_amd/csrc/host/align_gpu.cpp -- msa::progressive::gpu::alignmentKernel_GPU: the level kernel that drops in
// for cpu::alignmentKernel_CPU (/root/reference/src/alignment-cpu.cpp:32-183).  Same phases, same policy; the per-pair
// Talco_xdrop::Align_freq calls of a level (alignment-cpu.cpp:95-130) become twl_align_batch calls (include/twl_align.h).
// There is no CPU alignment path in this file: if the GPU library fails the run stops.
#include "align_gpu.hpp"

#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <iostream>
#include <map>
#include <string>
#include <tuple>

namespace msa {
namespace progressive {
namespace gpu {

LevelTotals g_totals;
static std::vector<int> g_devices;
const std::vector<int> &selectedDevices() { return g_devices; }

static std::future<std::pair<int, std::string>> g_initJob;   // (return code, twl_last_error() of the helper thread)

// Starts twl_init on a helper thread so that HIP start-up overlaps tree and FASTA reading; ensureInit() joins it.
void beginInit(Option *option)
{
    g_devices = option->gpuIdx.empty() ? std::vector<int>{0} : option->gpuIdx;
    g_initJob = std::async(std::launch::async, [] {
        const int rc = twl_init(g_devices.data(), (int)g_devices.size());
        return std::make_pair(rc, std::string(rc == TWL_OK ? "" : twl_last_error()));
    });
}

void ensureInit(Option *option)
{
    static bool done = false;
    if (done) return;
    if (!g_initJob.valid()) beginInit(option);
    const auto res = g_initJob.get();
    if (res.first != TWL_OK) { std::cerr << "ERROR: twl_init failed: " << res.second << '\n'; exit(1); }
    done = true;
}

twl_params baseParams(Params &param)          // == Talco_xdrop::Params(msa::Params&), TALCO-XDrop.cpp:36-53
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

// Level staging: the flat arrays of the C ABI ([pair][2][stride][P] etc.), kept for the whole run and only ever grown.  Each
// pair's profile and gap penalties are built in place in its slot (preparePair), so nothing is copied on the host and the pages
// are faulted in once, by the threads that fill them.
struct Staging {
    float *freq = nullptr, *gop = nullptr, *gex = nullptr;
    int8_t *aln = nullptr;
    size_t capFreq = 0, capGop = 0, capGex = 0, capAln = 0;
    template <class T> static void grow(T *&p, size_t &cap, size_t need)
    {
        if (need <= cap) return;
        std::free(p);
        cap = need + need / 8;
        p = static_cast<T *>(std::malloc(cap * sizeof(T)));
        if (!p) { std::cerr << "ERROR: out of host memory for the level staging (" << cap * sizeof(T) << " bytes)\n"; exit(1); }
    }
    void ensure(size_t n, size_t stride, size_t P)
    {
        grow(freq, capFreq, n * 2 * stride * P);
        grow(gop, capGop, n * 2 * stride);
        grow(gex, capGex, n * 2 * stride);
        grow(aln, capAln, n * 2 * stride);
    }
};
static Staging g_stage;

// Align the pairs `ids` (ascending indices into the level's slots) with one parameter set; results land in paths/errs.  The
// call covers the slot span [ids.front(), ids.back()]; slots in the span that are not in `ids` are passed with length 0, which the
// boundary answers with aln_len 0 without running them.
static void runBatch(const twl_params &tp, const std::vector<int> &ids, const std::vector<PairInputs> &in, int P, int stride,
                     std::vector<alnPath> &paths, std::vector<int16_t> &errs)
{
    if (ids.empty()) return;
    const double tStage = nowMs();
    const int first = ids.front(), n = ids.back() - first + 1;
    std::vector<int32_t> len(2 * (size_t)n, 0), num(2 * (size_t)n, 1), alnLen(n, 0);
    std::vector<int16_t> err(n, 0);
    for (int id : ids) {
        const PairInputs &pi = in[id];
        len[2 * (id - first)] = pi.lens.first; len[2 * (id - first) + 1] = pi.lens.second;
        num[2 * (id - first)] = pi.refNum; num[2 * (id - first) + 1] = pi.qryNum;
    }
    const size_t sl = (size_t)stride;
    int8_t *aln = g_stage.aln + (size_t)first * 2 * sl;
    const double tCall = nowMs();
    g_totals.stage_ms += tCall - tStage;
    int rc = twl_align_batch(&tp, n, stride, g_stage.freq + (size_t)first * 2 * sl * P, g_stage.gop + (size_t)first * 2 * sl,
                             g_stage.gex + (size_t)first * 2 * sl, len.data(), num.data(), aln, alnLen.data(), err.data());
    if (rc != TWL_OK) { std::cerr << "ERROR: twl_align_batch failed (" << rc << "): " << twl_last_error() << '\n'; exit(1); }
    g_totals.call_ms += nowMs() - tCall;
    for (int dev : g_devices) {
        twl_stats st{};
        if (twl_get_stats(dev, &st) == TWL_OK) { g_totals.band_cells += st.band_cells; g_totals.relaunched += (uint64_t)st.n_relaunched; g_totals.kernel_ms += st.kernel_ms; g_totals.total_ms += st.total_ms; }
    }
    g_totals.pairs += ids.size();
    for (int id : ids) {
        const int t = id - first;
        errs[id] = err[t];
        paths[id].assign(aln + (size_t)t * 2 * sl, aln + (size_t)t * 2 * sl + (err[t] == 0 ? alnLen[t] : 0));
    }
}

void alignmentKernel_GPU(Tree *, NodePairVec &nodes, SequenceDB *database, Option *option, Params &param)
{
    if (option->cpuOnly) { std::cerr << "ERROR: --cpu-only is not available: this build has no CPU alignment path.\n"; exit(1); }
    ensureInit(option);
    const int n = (int)nodes.size();
    const int P = param.matrixSize + 1;
    std::vector<PairInputs> in(n);
    const double tPrep = nowMs();
    const LevelTotals before = g_totals;
    // wide levels: one pair per thread; narrow levels (upper tree): pairs in turn, the helpers' own column/sequence loops fan out
    const bool acrossPairs = n >= omp_get_max_threads();
    int stride = 1;
    for (auto &pr : nodes) stride = std::max({stride, pr.first->getAlnLen(database->currentTask), pr.second->getAlnLen(database->currentTask)});
    g_stage.ensure((size_t)n, (size_t)stride, (size_t)P);
    const size_t sl = (size_t)stride;
#pragma omp parallel for schedule(dynamic, 1) if (acrossPairs)
    for (int i = 0; i < n; ++i)                                                                 // alignment-cpu.cpp:50-93
        preparePair(nodes[i], database, option, param, in[i], g_stage.freq + (size_t)i * 2 * sl * P, g_stage.gop + (size_t)i * 2 * sl,
                    g_stage.gex + (size_t)i * 2 * sl, stride);

    g_totals.prepare_ms += nowMs() - tPrep;

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
    runBatch(tp, plain, in, P, stride, paths, errs);
    twl_params tz = tp;
    tz.gap_char = 0;
    runBatch(tz, zeroGap, in, P, stride, paths, errs);

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
            runBatch(tr, std::vector<int>{i}, in, P, stride, paths, errs);
        }
    }

    std::vector<char> deferred(n, 0);
    const double tFin = nowMs();
#pragma omp parallel for schedule(dynamic, 1) if (acrossPairs)
    for (int i = 0; i < n; ++i) {                                                              // alignment-cpu.cpp:136-175
        // low-quality singleton rule (:136-144): such a pair is deferred whatever the DP said
        deferred[i] = (database->currentTask == 0 && (in[i].refNum == 1 || in[i].qryNum == 1) && (in[i].lowQ_r || in[i].lowQ_q)) ? 1 : 0;
        finishPair(nodes[i], database, option, param, in[i], paths[i]);
        in[i] = PairInputs();                                                                   // release the profile buffers early
    }
    g_totals.finish_ms += nowMs() - tFin;
    for (int i = 0; i < n; ++i)
        if (deferred[i]) fallbackPairs.push_back(i);
    if (!fallbackPairs.empty()) alignment_helper::fallback2cpu(fallbackPairs, nodes, database, option);
    if (option->printDetail)
        std::cerr << "  phases (ms): prepare " << g_totals.prepare_ms - before.prepare_ms << " stage " << g_totals.stage_ms - before.stage_ms << " call "
                  << g_totals.call_ms - before.call_ms << " (kernel " << g_totals.kernel_ms - before.kernel_ms << ") finish " << g_totals.finish_ms - before.finish_ms
                  << " whole " << nowMs() - tPrep << '\n';
}

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

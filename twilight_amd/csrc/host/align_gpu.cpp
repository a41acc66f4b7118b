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
#include <cstring>
#include <future>
#include <iostream>
#include <map>
#include <string>
#include <tuple>

namespace msa {
namespace progressive {
namespace gpu {

static std::vector<int> g_devices;

// Page-locked when the library can give it (copies at link speed), else plain memory.  Pinned blocks are not returned at process exit:
// the staging objects are process-wide statics and the HIP runtime may be gone before their destructors run.
char *RunCtx::Raw::get(size_t n)
{
    if (n > cap) {
        if (pinned) twl_host_free(p); else free(p);
        cap = n + n / 8 + 64;
        p = static_cast<char *>(twl_host_alloc(cap));
        pinned = (p != nullptr);
        if (!p) p = static_cast<char *>(malloc(cap));
        if (!p) { std::cerr << "ERROR: out of host memory for the level staging (" << cap << " bytes)\n"; exit(1); }
    }
    return p;
}
RunCtx::Raw::~Raw() { if (!pinned) free(p); }

RunCtx::~RunCtx()
{
    for (twl_store *st : stores) twl_store_destroy(st);
}
RunCtx &ctxOf(SequenceDB *database)
{
    if (!database->gpuCtx) {
        database->gpuCtx = new RunCtx();
        database->gpuCtxFree = [](void *p) { delete static_cast<RunCtx *>(p); };
    }
    return *static_cast<RunCtx *>(database->gpuCtx);
}
void setShard(SequenceDB *database, const Shard &shard)
{
    ctxOf(database).shard = shard;
    database->ownedPrefix = nullptr;
    if (shard.world > 1)      // (callers that run the host-staged kernel clear it again: capi.cpp)
        database->ownedPrefix = [database](Tree *T, std::vector<NodePairVec> &levels, Option *option, Params &param) { return ownedPrefix(T, levels, database, option, param); };
}
const std::vector<LevelRecord> &levelRecords(SequenceDB *database) { return ctxOf(database).levels; }
const LevelTotals &runTotals(SequenceDB *database) { return ctxOf(database).totals; }

std::vector<int> dealPairs(const std::vector<long long> &cost, const std::vector<char> &takesPart, int parts)
{
    const int n = (int)cost.size();
    std::vector<int> owner(n, 0);
    if (parts <= 1) return owner;
    std::vector<int> order;
    for (int i = 0; i < n; ++i) if (takesPart[i]) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
    std::vector<long long> load(parts, 0);
    for (int i : order) {
        const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        owner[i] = d;
        load[d] += std::max<long long>(cost[i], 1);
    }
    return owner;
}

// Block layout per rank: header {band cells u64, relaunched u64, kernel ms f64, reserved} then, for the rank's pairs in ascending
// order, {path length i32, errorType i32, path bytes padded to 8}: rows as long as their paths.  Every rank knows every rank's pair
// list (the deal is deterministic), so only the blocks travel -- after an 8-byte all-gather of the block sizes, because the
// collective wants equal blocks and the paths are about half as long as the bound 2 * seq_len the rows used to be padded to.
// The blocks live in grow-only staging that is never zero-filled (page-locked when the library can give it).
void exchangePaths(RunCtx &ctx, const std::vector<int> &owner, const std::vector<char> &takesPart, int pathCap, std::vector<alnPath> &paths,
                   std::vector<int16_t> &errs, LevelRecord &rec)
{
    const Shard &sh = ctx.shard;
    if (!sh.exchange && !sh.rccl) {          // (a 1-rank "world" with an exchange function still goes through it: lets one GPU cover the collective)
        if (sh.world > 1) { std::cerr << "ERROR: sharded run without an exchange function.\n"; exit(1); }
        return;
    }
    // the all-gather of host blocks: the caller's callback, or the library's own RCCL communicator (blocks staged through its device buffers)
    auto gather = [&](const void *send, int64_t bytes, void *recv) -> int {
        if (sh.rccl) { const int rc = twl_comm_all_gather_host(g_devices[0], send, recv, bytes); if (rc != TWL_OK) std::cerr << "ERROR: " << twl_last_error() << '\n'; return rc; }
        return sh.exchange(sh.user, send, bytes, recv);
    };
    static RunCtx::Raw sendStage, recvStage;      // (one run aligns at a time per process)
    constexpr uint64_t kBlockMagic = 0x54574C50ull << 32;      // "TWLP"
    const double t0 = nowMs();
    const int n = (int)owner.size();
    std::vector<std::vector<int>> mine(sh.world);
    for (int i = 0; i < n; ++i) if (takesPart[i]) mine[owner[i]].push_back(i);
    const size_t head = 32;
    auto rowBytes = [](size_t len) { return 8 + ((len + 7) & ~(size_t)7); };
    size_t myBytes = head;
    for (int i : mine[sh.rank]) {
        if ((int)paths[i].size() > pathCap) { std::cerr << "ERROR: path longer than the exchange row.\n"; exit(1); }
        myBytes += rowBytes(paths[i].size());
    }
    // block sizes first (8 bytes per rank)
    std::vector<int64_t> sizes((size_t)sh.world, 0);
    {
        const int64_t mineSz = (int64_t)myBytes;
        const int rc0 = gather(&mineSz, (int64_t)sizeof(int64_t), sizes.data());
        if (rc0 != 0) { std::cerr << "ERROR: exchange of the block sizes failed (" << rc0 << ").\n"; exit(1); }
    }
    const size_t blockBytes = (size_t)*std::max_element(sizes.begin(), sizes.end());
    char *send = sendStage.get(blockBytes), *recv = recvStage.get(blockBytes * (size_t)sh.world);
    {
        uint64_t h[4] = {rec.band_cells, rec.relaunched, 0, kBlockMagic | (uint64_t)(uint32_t)ctx.levels.size()};      // (magic, level number)
        memcpy(&h[2], &rec.kernel_ms, sizeof(double));
        memcpy(send, h, sizeof h);
        size_t at = head;
        for (int i : mine[sh.rank]) {
            const int32_t len = (int32_t)paths[i].size(), e = errs[i];
            memcpy(&send[at], &len, 4); memcpy(&send[at + 4], &e, 4);
            if (len) memcpy(&send[at + 8], paths[i].data(), (size_t)len);
            at += rowBytes((size_t)len);
        }
    }
    const int rc = gather(send, (int64_t)blockBytes, recv);
    if (rc != 0) { std::cerr << "ERROR: exchange of the level's paths failed (" << rc << ").\n"; exit(1); }
    rec.band_cells = 0; rec.relaunched = 0;
    double kmax = 0;
    for (int r = 0; r < sh.world; ++r) {
        const char *blk = &recv[blockBytes * (size_t)r];
        uint64_t h[4]; memcpy(h, blk, sizeof h);
        // the receive staging is never zero-filled: a block that is not this level's (a collective that failed without saying so) must not parse
        if (h[3] != (kBlockMagic | (uint64_t)(uint32_t)ctx.levels.size())) { std::cerr << "ERROR: path block of rank " << r << " does not belong to this level.\n"; exit(1); }
        double km; memcpy(&km, &h[2], sizeof(double));
        rec.band_cells += h[0]; rec.relaunched += h[1]; kmax = std::max(kmax, km);      // the ranks ran concurrently
        size_t at = head;
        for (int i : mine[r]) {
            int32_t len, e; memcpy(&len, blk + at, 4); memcpy(&e, blk + at + 4, 4);
            if (len < 0 || at + rowBytes((size_t)len) > (size_t)sizes[r]) { std::cerr << "ERROR: malformed path block from rank " << r << ".\n"; exit(1); }
            if (r != sh.rank) { errs[i] = (int16_t)e; paths[i].assign(blk + at + 8, blk + at + 8 + len); }
            at += rowBytes((size_t)len);
        }
    }
    rec.kernel_ms = kmax;
    rec.exchange_ms += nowMs() - t0;
}

void exchangeFinalPaths(RunCtx &ctx, twl_store *store, int device, const twl_params &tp, const std::vector<int> &owner, const std::vector<char> &takesPart,
                        const std::vector<int32_t> &bound, int pathStride, std::vector<uint8_t> &fromDp, std::vector<int32_t> &dpLen, std::vector<int16_t> &errs,
                        LevelRecord &rec)
{
    const Shard &sh = ctx.shard;
    const double t0 = nowMs();
    auto die = [](const char *what, int rc) { std::cerr << "ERROR: " << what << " failed (" << rc << "): " << twl_last_error() << '\n'; exit(1); };
    constexpr uint64_t kBlockMagic = 0x54574C44ull << 32;      // "TWLD"
    const int n = (int)owner.size(), world = sh.world;
    std::vector<std::vector<int>> mine(world);
    for (int i = 0; i < n; ++i) if (takesPart[i]) mine[owner[i]].push_back(i);
    auto pad8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    // block of rank r: {band cells, relaunched, kernel ms, magic | level} + {length, errorType} per pair of r + the paths, each padded to 8
    std::vector<size_t> hdr(world);
    size_t blockMax = 0, hdrMax = 0;
    for (int r = 0; r < world; ++r) {
        hdr[r] = 32 + 8 * mine[r].size();
        size_t cap = hdr[r];
        for (int i : mine[r]) cap += pad8((size_t)bound[i]);
        blockMax = std::max(blockMax, cap); hdrMax = std::max(hdrMax, hdr[r]);
    }
    blockMax = (blockMax + 255) & ~(size_t)255;
    int rc = twl_level_restore(store, &tp, 0, nullptr, pathStride, nullptr);      // (fixes the row pitch of the path buffer if no pair was restored)
    if (rc != TWL_OK) die("twl_level_restore", rc);
    void *send = nullptr, *recv = nullptr;
    if ((rc = twl_level_exchange_buffers(store, (int64_t)blockMax, (int64_t)(blockMax * (size_t)world), &send, &recv)) != TWL_OK) die("twl_level_exchange_buffers", rc);
    const std::vector<int> &me = mine[sh.rank];
    {
        std::vector<char> h(hdr[sh.rank]);
        uint64_t h4[4] = {rec.band_cells, rec.relaunched, 0, kBlockMagic | (uint64_t)(uint32_t)ctx.levels.size()};
        memcpy(&h4[2], &rec.kernel_ms, sizeof(double));
        memcpy(h.data(), h4, sizeof h4);
        std::vector<int32_t> prs, lens;
        std::vector<uint8_t> where;
        std::vector<int64_t> off;
        size_t at = hdr[sh.rank];
        for (size_t t = 0; t < me.size(); ++t) {
            const int i = me[t];
            const int32_t len = (errs[i] == 0 && fromDp[i]) ? dpLen[i] : 0, e = errs[i];
            memcpy(&h[32 + 8 * t], &len, 4); memcpy(&h[32 + 8 * t + 4], &e, 4);
            if (len > 0) { prs.push_back(i); lens.push_back(len); where.push_back(fromDp[i]); off.push_back((int64_t)at); at += pad8((size_t)len); }
        }
        if (at > blockMax) { std::cerr << "ERROR: final paths longer than their bound.\n"; exit(1); }
        if ((rc = twl_copy_to_device(device, send, h.data(), h.size())) != TWL_OK) die("twl_copy_to_device", rc);
        if ((rc = twl_level_paths_to_block(store, (int32_t)prs.size(), prs.data(), lens.data(), where.data(), send, off.data())) != TWL_OK) die("twl_level_paths_to_block", rc);
    }
    const int xrc = sh.rccl ? twl_comm_all_gather(device, send, recv, (int64_t)blockMax)      // ncclAllGather on the library's stream: HBM to HBM over xGMI
                            : sh.exchangeDev(sh.userDev, send, (int64_t)blockMax, recv);
    if (xrc != 0) { std::cerr << "ERROR: device exchange of the level's paths failed (" << xrc << ").\n"; exit(1); }
    std::vector<char> hb(hdrMax * (size_t)world);
    if ((rc = twl_copy_rows_from_device(device, hb.data(), hdrMax, recv, blockMax, hdrMax, (uint64_t)world)) != TWL_OK) die("twl_copy_rows_from_device", rc);
    rec.band_cells = 0; rec.relaunched = 0;
    double kmax = 0;
    std::vector<int32_t> prs, lens;
    std::vector<int64_t> off;
    for (int r = 0; r < world; ++r) {
        const char *blk = &hb[hdrMax * (size_t)r];
        uint64_t h4[4]; memcpy(h4, blk, sizeof h4);
        if (h4[3] != (kBlockMagic | (uint64_t)(uint32_t)ctx.levels.size())) { std::cerr << "ERROR: path block of rank " << r << " does not belong to this level.\n"; exit(1); }
        double km; memcpy(&km, &h4[2], sizeof(double));
        rec.band_cells += h4[0]; rec.relaunched += h4[1]; kmax = std::max(kmax, km);
        size_t at = hdr[r];
        for (size_t t = 0; t < mine[r].size(); ++t) {
            const int i = mine[r][t];
            int32_t len, e; memcpy(&len, blk + 32 + 8 * t, 4); memcpy(&e, blk + 32 + 8 * t + 4, 4);
            if (len < 0 || len > bound[i] || at + pad8((size_t)len) > blockMax) { std::cerr << "ERROR: malformed path block from rank " << r << ".\n"; exit(1); }
            if (r != sh.rank) {
                errs[i] = (int16_t)e; dpLen[i] = len; fromDp[i] = len > 0 ? 2 : 0;
                if (len > 0) { prs.push_back(i); lens.push_back(len); off.push_back((int64_t)(blockMax * (size_t)r + at)); }
            }
            at += pad8((size_t)len);
        }
    }
    if ((rc = twl_level_paths_from_block(store, (int32_t)prs.size(), prs.data(), lens.data(), recv, off.data())) != TWL_OK) die("twl_level_paths_from_block", rc);
    rec.kernel_ms = kmax;
    rec.exchange_ms += nowMs() - t0;
}

const std::vector<int> &selectedDevices() { return g_devices; }

void ensureDevicesUp(Option *option) { ensureInit(option); }

int initRcclShard(SequenceDB *database, Option *option, int rank, int world, const void *id128)
{
    ensureInit(option);
    if (g_devices.size() != 1) { std::cerr << "ERROR: a sharded run takes one device per process.\n"; exit(1); }
    const int rc = twl_comm_init(g_devices[0], rank, world, id128);
    if (rc != TWL_OK) return rc;            // (twl_last_error() says why; the CLI ends the run, a caller of the C ABI may shard through its own collective instead)
    Shard sh;
    sh.rank = rank; sh.world = world; sh.rccl = true;
    setShard(database, sh);
    return TWL_OK;
}

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
static void runBatch(RunCtx &ctx, LevelRecord &rec, const twl_params &tp, const std::vector<int> &ids, const std::vector<PairInputs> &in, int P, int stride,
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
#ifdef TWL_DEV      // development builds: the inputs of up to TWL_DUMP_BATCH_N calls as raw arrays (tools/load_dump.py), to replay a call on the CPU checker
    if (const char *dump = getenv("TWL_DUMP_BATCH")) {
        static int left = getenv("TWL_DUMP_BATCH_N") ? atoi(getenv("TWL_DUMP_BATCH_N")) : 4;
        const int minStride = getenv("TWL_DUMP_BATCH_MIN") ? atoi(getenv("TWL_DUMP_BATCH_MIN")) : 0;
        if (left > 0 && stride >= minStride) {
            if (FILE *f = fopen(dump, "ab")) {
                --left;
                const int32_t hd[8] = {0x7477626c, P, n, stride, tp.xdrop, tp.flen, tp.marker, 0};
                fwrite(hd, sizeof hd, 1, f);
                const float gp[4] = {tp.gap_open, tp.gap_extend, tp.gap_char, 0.f};
                fwrite(gp, sizeof gp, 1, f);
                fwrite(tp.matrix, sizeof(float), (size_t)(P - 1) * (P - 1), f);
                fwrite(len.data(), sizeof(int32_t), len.size(), f);
                fwrite(num.data(), sizeof(int32_t), num.size(), f);
                fwrite(g_stage.freq + (size_t)first * 2 * sl * P, sizeof(float), (size_t)n * 2 * sl * P, f);
                fwrite(g_stage.gop + (size_t)first * 2 * sl, sizeof(float), (size_t)n * 2 * sl, f);
                fwrite(g_stage.gex + (size_t)first * 2 * sl, sizeof(float), (size_t)n * 2 * sl, f);
                fclose(f);
            }
        }
    }
#endif
    const double tCall = nowMs();
    ctx.totals.stage_ms += tCall - tStage;
    int rc = twl_align_batch(&tp, n, stride, g_stage.freq + (size_t)first * 2 * sl * P, g_stage.gop + (size_t)first * 2 * sl,
                             g_stage.gex + (size_t)first * 2 * sl, len.data(), num.data(), aln, alnLen.data(), err.data());
    if (rc != TWL_OK) { std::cerr << "ERROR: twl_align_batch failed (" << rc << "): " << twl_last_error() << '\n'; exit(1); }
    ctx.totals.call_ms += nowMs() - tCall;
    double kms = 0, tms = 0;
    for (int dev : g_devices) {      // the devices of one call run concurrently: cells add up, times do not
        twl_stats st{};
        if (twl_get_stats(dev, &st) == TWL_OK) { ctx.totals.nominal_cells += st.nominal_cells; rec.band_cells += st.band_cells; rec.relaunched += (uint64_t)st.n_relaunched; kms = std::max(kms, st.kernel_ms); tms = std::max(tms, st.total_ms);
            if (rec.matrix_mode < 0) { rec.matrix_mode = st.matrix_mode; rec.speculative = st.speculative; memcpy(rec.kernel, st.kernel, sizeof rec.kernel); }
            rec.mt_predicted += st.mt_tiles_predicted; rec.mt_inline += st.mt_tiles_inline; }
    }
    rec.kernel_ms += kms;
    ctx.totals.total_ms += tms;
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
    RunCtx &ctx = ctxOf(database);
    LevelRecord rec;
    rec.pairs = (int32_t)nodes.size();
    rec.task = database->currentTask;
    const int n = (int)nodes.size();
    const int P = param.matrixSize + 1;
    std::vector<PairInputs> in(n);
    const double tPrep = nowMs();
    const LevelTotals before = ctx.totals;
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

    ctx.totals.prepare_ms += nowMs() - tPrep;

    std::vector<alnPath> paths(n);
    std::vector<int16_t> errs(n, 0);
    // pairs that go to the DP, grouped by gapCharScore (alignment-cpu.cpp:88); with several processes each aligns the pairs dealt to it
    std::vector<char> takesPart(n, 0);
    std::vector<long long> cost(n, 0);
    for (int i = 0; i < n; ++i) {
        if (in[i].refLen == 0) paths[i].assign(in[i].qryLen, 1);                              // :89-90
        if (in[i].qryLen == 0) paths[i].insert(paths[i].end(), in[i].refLen, 2);
        if (!paths[i].empty() || in[i].lowQ_r || in[i].lowQ_q) continue;                       // :93,95
        takesPart[i] = 1;
        cost[i] = (long long)in[i].lens.first + in[i].lens.second;
    }
    const std::vector<int> owner = dealPairs(cost, takesPart, ctx.shard.world);
    std::vector<int> plain, zeroGap;
    auto zeroGapPair = [&](int i) { return database->currentTask == 1 || database->currentTask == 2 || in[i].refNum > 10000 || in[i].qryNum > 10000; };
    for (int i = 0; i < n; ++i)
        if (takesPart[i] && owner[i] == ctx.shard.rank) (zeroGapPair(i) ? zeroGap : plain).push_back(i);
    twl_params tp = baseParams(param);
    runBatch(ctx, rec, tp, plain, in, P, stride, paths, errs);
    twl_params tz = tp;
    tz.gap_char = 0;
    runBatch(ctx, rec, tz, zeroGap, in, P, stride, paths, errs);

    // alignment-cpu.cpp:116-129: in the later tasks a failed pair is retried with a larger X-drop / band limit (by the rank that owns it)
    if (database->currentTask != 0) {
        for (int i = 0; i < n; ++i) {
            if (!takesPart[i] || owner[i] != ctx.shard.rank || errs[i] == 0) continue;
            twl_params tr = zeroGapPair(i) ? tz : tp;
            const int minLen = std::min(in[i].lens.first, in[i].lens.second);
            while (errs[i] != 0) {
                if (errs[i] == 3) { std::cout << "There might be some bugs in the code!\n"; exit(1); }
                if (errs[i] == 2) tr.flen = std::min(static_cast<int32_t>(tr.flen * 1.2) << 1, minLen);
                else { tr.xdrop = static_cast<int32_t>(tr.xdrop * 2); tr.flen = std::min(static_cast<int32_t>(tr.xdrop * 4) << 1, minLen); }
                if (option->printDetail) std::cout << "Retry pair No. " << i << "\txdrop " << tr.xdrop << " flen " << tr.flen << '\n';
                runBatch(ctx, rec, tr, std::vector<int>{i}, in, P, stride, paths, errs);
            }
        }
    }
    exchangePaths(ctx, owner, takesPart, 2 * stride, paths, errs, rec);

    // alignment-cpu.cpp:108-115: task 0 defers a failed pair
    std::vector<int> fallbackPairs;
    for (int i = 0; i < n; ++i) {
        if (errs[i] == 0) continue;
        if (errs[i] == 3) { std::cout << "There might be some bugs in the code!\n"; exit(1); }
        paths[i].clear();
        fallbackPairs.push_back(i);
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
    ctx.totals.finish_ms += nowMs() - tFin;
    for (int i = 0; i < n; ++i)
        if (deferred[i]) fallbackPairs.push_back(i);
    if (!fallbackPairs.empty()) alignment_helper::fallback2cpu(fallbackPairs, nodes, database, option);
    rec.level_ms = nowMs() - tPrep;
    ctx.totals.pairs += (uint64_t)n; ctx.totals.band_cells += rec.band_cells; ctx.totals.relaunched += rec.relaunched; ctx.totals.kernel_ms += rec.kernel_ms; ctx.totals.exchange_ms += rec.exchange_ms;
    ctx.levels.push_back(rec);
    if (option->printDetail)
        std::cerr << "  phases (ms): prepare " << ctx.totals.prepare_ms - before.prepare_ms << " stage " << ctx.totals.stage_ms - before.stage_ms << " call "
                  << ctx.totals.call_ms - before.call_ms << " (kernel " << ctx.totals.kernel_ms - before.kernel_ms << ") finish " << ctx.totals.finish_ms - before.finish_ms
                  << " whole " << nowMs() - tPrep << '\n';
}

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

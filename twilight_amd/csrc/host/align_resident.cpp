// twilight_amd/csrc/host/align_resident.cpp -- msa::progressive::gpu::alignmentKernel_Resident: the level kernel with the
// sequences resident in HBM (include/twl_level.h).  Same phases and policy as cpu::alignmentKernel_CPU
// (/root/reference/src/alignment-cpu.cpp:32-183); what moves to the device besides the DP is calculateProfile,
// getConsensus, removeGappyColumns, calculatePSGP (before) and updateFrequency + the row rewriting of updateAlignment (after).
// The host keeps the policy: empty sides, low-quality singletons, retry/defer, addGappyColumnsBack, Node bookkeeping.
//
// Differences from the host-staged kernel that do not change the result:
//  * groups of > 1000 sequences are not compressed into one negative id (alignment-helper.cpp:479-500): that device only saves the
//    reference per-level row rewriting, which here is a sub-millisecond kernel; every row is simply kept up to date.
//  * the deferred pass (currentTask 1, /root/reference/src/progressive.cpp:270-298) runs here too since round 4: the rows and the cached
//    profiles of the root and of the deferred nodes stay in HBM, every pair takes gapCharScore 0 and a failed pair is retried with the
//    larger X-drop / band limit of /root/reference/src/alignment-cpu.cpp:116-129 (per pair: twl_level_align with a one-pair mask).  Its
//    profiles cannot be batched: each is aligned to the root the previous one has just been merged into (:283-285).  Rows come back to
//    the host after it (progressive.cpp of this directory); currentTask 2 and a run whose rows are on the host already take alignmentKernel_GPU.
//
// Several devices (phase 2) without any collective: every device holds a replica of the store.  Profile building and the
// write-back are ~1 % of a level's device time, so EVERY device runs them for ALL pairs of the level (the replicas stay identical by
// construction, cached profiles included); only the DP -- the 99 % -- is sharded: each device aligns its share of the pairs
// (twl_level_align with a mask, longest-processing-time deal), the paths meet on the host, and every device commits all of them.
// Per level only paths cross PCIe, nothing crosses xGMI.  --test-virtual-devices k (tests) runs k replicas on the first device.
#include "align_gpu.hpp"

#include <cstring>
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

namespace msa {
namespace progressive {
namespace gpu {

namespace {

// the per-run state (store replicas, cache ids, totals) lives in RunCtx (align_gpu.hpp), reached through the SequenceDB
inline twl_store *firstStore(RunCtx &ctx) { return ctx.stores.empty() ? nullptr : ctx.stores[0]; }

// Grow-only, never zero-filled staging shared by the runs of a process (one run aligns at a time): a level of 3 000 pairs x 10 kbp
// moves ~70 MB per array and first-touch page faults cost more than the copies.
std::vector<RunCtx::Raw> g_alnStage;      // per replica: paths as the DP wrote them
RunCtx::Raw g_finalStage, g_infoStage;    // final paths as the commit reads them; column info

void die(const char *what, int rc)
{
    std::cerr << "ERROR: " << what << " failed (" << rc << "): " << twl_last_error() << '\n';
    exit(1);
}

void createStore(RunCtx &ctx, SequenceDB *db, Option *option)
{
    const int n = (int)db->sequences.size();
    std::vector<const char *> rows(n);
    std::vector<int32_t> lens(n);
    for (int i = 0; i < n; ++i) {
        auto *s = db->sequences[i];
        if (s->id != i) { std::cerr << "ERROR: sequence ids are not dense.\n"; exit(1); }
        rows[i] = s->alnStorage[s->storage];
        lens[i] = s->len;
    }
    ctx.storeDev = selectedDevices();
    if (option->testVirtualDevices > 0) ctx.storeDev.assign((size_t)option->testVirtualDevices, selectedDevices()[0]);
    ctx.stores.assign(ctx.storeDev.size(), nullptr);
    std::vector<std::thread> th;
    std::vector<std::pair<int, std::string>> res(ctx.storeDev.size(), {TWL_OK, ""});
    for (size_t d = 0; d < ctx.storeDev.size(); ++d)
        th.emplace_back([&, d] {
            const int rc = twl_store_create(ctx.storeDev[d], option->type, n, rows.data(), lens.data(), &ctx.stores[d]);
            if (rc != TWL_OK) res[d] = {rc, twl_last_error()};
        });
    for (auto &t : th) t.join();
    for (auto &r : res)
        if (r.first != TWL_OK) { std::cerr << "ERROR: twl_store_create failed (" << r.first << "): " << r.second << '\n'; exit(1); }
}

// Runs fn(replica index) for every replica on its own host thread; stops the run on the first library error.
template <class F>
void onAllStores(RunCtx &ctx, const char *what, F fn)
{
    const size_t nd = ctx.stores.size();
    std::vector<std::pair<int, std::string>> res(nd, {TWL_OK, ""});
    if (nd == 1) { const int rc = fn(0); if (rc != TWL_OK) res[0] = {rc, twl_last_error()}; }
    else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < nd; ++d)
            th.emplace_back([&, d] { const int rc = fn((int)d); if (rc != TWL_OK) res[d] = {rc, twl_last_error()}; });
        for (auto &t : th) t.join();
    }
    for (auto &r : res)
        if (r.first != TWL_OK) { std::cerr << "ERROR: " << what << " failed (" << r.first << "): " << r.second << '\n'; exit(1); }
}

// End of the main pass: current rows back into SequenceInfo::alnStorage, cached profiles of the nodes that go on (the root and the
// deferred nodes) back into Node::msaFreq, store released.
void materialise(Tree *T, SequenceDB *db, Option *option)
{
    RunCtx &ctx = ctxOf(db);
    if (!firstStore(ctx)) return;
    const double t0 = nowMs();
    const int n = (int)db->sequences.size();
    std::vector<int32_t> lens(n);
    int rc = twl_store_read_rows(firstStore(ctx), nullptr, lens.data());
    if (rc != TWL_OK) die("twl_store_read_rows", rc);
    // One block for all rows, laid out like the device's gathered buffer (row i at the prefix sum of the lengths), so the library
    // copies straight into it; the second half backs the sequences' other buffer (untouched until a later pass writes there).
    std::vector<char *> rows(n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)lens[i];
    free(db->rowArena);
    db->rowArena = static_cast<char *>(malloc(2 * total + 1));
    if (!db->rowArena) { std::cerr << "ERROR: out of host memory for the final rows.\n"; exit(1); }
    size_t at = 0;
    for (int i = 0; i < n; ++i) {
        auto *s = db->sequences[i];
        if (!s->borrowed) { free(s->alnStorage[0]); free(s->alnStorage[1]); }
        s->borrowed = true;
        s->alnStorage[s->storage] = db->rowArena + at;
        s->alnStorage[!s->storage] = db->rowArena + total + at;
        s->memLen = lens[i];
        rows[i] = s->alnStorage[s->storage];
        at += (size_t)lens[i];
    }
    rc = twl_store_read_rows(firstStore(ctx), rows.data(), lens.data());
    if (rc != TWL_OK) die("twl_store_read_rows", rc);
    for (int i = 0; i < n; ++i) db->sequences[i]->len = lens[i];
    const int P = (option->type == 'n') ? 6 : 22;
    std::vector<Node *> keep{T->root};
    keep.insert(keep.end(), db->fallback_nodes.begin(), db->fallback_nodes.end());
    for (Node *nd : keep) {
        if (nd->cacheId < 0) continue;
        int32_t len = 0;
        if ((rc = twl_store_read_cache(firstStore(ctx), nd->cacheId, nullptr, &len)) != TWL_OK) die("twl_store_read_cache", rc);
        std::vector<float> flat((size_t)len * P);
        if ((rc = twl_store_read_cache(firstStore(ctx), nd->cacheId, flat.data(), &len)) != TWL_OK) die("twl_store_read_cache", rc);
        nd->msaFreq.assign(len, std::vector<float>(P));
        for (int t = 0; t < len; ++t) std::copy(&flat[(size_t)t * P], &flat[(size_t)t * P] + P, nd->msaFreq[t].begin());
        nd->cacheId = -1;
    }
    for (twl_store *st : ctx.stores) twl_store_destroy(st);
    ctx.stores.clear();
    ctx.finished = true;
    if (option->printDetail) std::cerr << "Rows back on the host in " << nowMs() - t0 << " ms\n";
}

struct PairState {
    int32_t refLen, qryLen, refNum, qryNum;
    IntPair lens;
    bool lowQ_r, lowQ_q;
    std::pair<IntPairVec, IntPairVec> gappy;
    stringPair consensus;
};

void runsAndConsensus(const uint8_t *info, int len, bool removal, const char *letters, IntPairVec &runs, std::string &cons)
{
    cons.resize(len);
    int start = -1;
    for (int t = 0; t < len; ++t) {
        cons[t] = letters[info[t] & 0x7f];
        if (!removal) continue;
        if (info[t] & 0x80) { if (start < 0) start = t; }
        else if (start >= 0) { runs.push_back({start, t - start}); start = -1; }
    }
    if (removal && start >= 0) runs.push_back({start, len - start});
}

}  // namespace

void uploadSequences(SequenceDB *database, Option *option)
{
    RunCtx &ctx = ctxOf(database);
    if (firstStore(ctx) || ctx.finished) return;
    ensureInit(option);
    const double t0 = nowMs();
    createStore(ctx, database, option);
    // the rows live in HBM from here on; the host copies (two buffers per sequence) come back as one arena when something on the host
    // needs them (materialise): a resident run that waits for its turn holds names and lengths only
    for (auto *s : database->sequences) {
        if (!s->borrowed) { free(s->alnStorage[0]); free(s->alnStorage[1]); }
        s->alnStorage[0] = s->alnStorage[1] = nullptr;
        s->borrowed = true;
        s->memLen = 0;
    }
    if (option->printDetail) std::cerr << "Sequences resident on " << ctx.stores.size() << " device replica(s) in " << nowMs() - t0 << " ms\n";
    database->afterMainPass = [database, option](Tree *tree) { materialise(tree, database, option); };
    database->residentDeferred = true;      // the level kernel that set this up also takes the deferred pass with the rows in HBM
}

void downloadRows(SequenceDB *database, Tree *T)
{
    if (database->afterMainPass) { database->afterMainPass(T); database->afterMainPass = nullptr; }
}

void alignmentKernel_Resident(Tree *T, NodePairVec &nodes, SequenceDB *database, Option *option, Params &param)
{
    RunCtx &ctx = ctxOf(database);
    const int task = database->currentTask;
    if ((task != 0 && task != 1) || ctx.finished) {      // merging sub-alignments, or the rows are on the host already: host-staged kernel
        if (firstStore(ctx)) materialise(T, database, option);
        alignmentKernel_GPU(T, nodes, database, option, param);
        return;
    }
    if (option->cpuOnly) { std::cerr << "ERROR: --cpu-only is not available: this build has no CPU alignment path.\n"; exit(1); }
    ensureInit(option);
    uploadSequences(database, option);
    if (ctx.shard.world > 1 && ctx.stores.size() > 1) { std::cerr << "ERROR: several processes with several device replicas each are not supported.\n"; exit(1); }
    LevelRecord rec;
    rec.pairs = (int32_t)nodes.size();
    rec.task = task;
    const LevelTotals before = ctx.totals;
    const double tPrep = nowMs();
    const int n = (int)nodes.size();
    static const char bases[] = {'A', 'C', 'G', 'T', 'N'};
    static const char acids[] = {'A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M', 'N', 'P', 'Q', 'R', 'S', 'T', 'V', 'W', 'Y', 'X'};
    const char *letters = (option->type == 'n') ? bases : acids;

    // ---- side descriptors (what calculateProfile reads from the two nodes, alignment-helper.cpp:8-40) ----
    std::vector<twl_side> sides(2 * (size_t)n);
    std::vector<PairState> ps(n);
    size_t nMembers = 0;
    int stride = 1;
    // (what is read from the nodes and sequences of a pair -- cache misses, mostly -- in parallel; the running member offsets and the
    // ids of newly cached profiles, which depend on the order of the pairs, in a serial pass behind it)
#pragma omp parallel for schedule(static) if (n >= 512)
    for (int i = 0; i < n; ++i) {
        Node *nd[2] = {nodes[i].first, nodes[i].second};
        PairState &s = ps[i];
        s.refLen = nd[0]->getAlnLen(0); s.qryLen = nd[1]->getAlnLen(0);
        s.refNum = nd[0]->getAlnNum(0); s.qryNum = nd[1]->getAlnNum(0);
        for (int sd = 0; sd < 2; ++sd) {
            twl_side &x = sides[2 * (size_t)i + sd];
            x.n_members = (int32_t)nd[sd]->seqsIncluded.size();
            x.len = sd ? s.qryLen : s.refLen;
            x.num = sd ? s.qryNum : s.refNum;
            x.weight = nd[sd]->alnWeight;
            x.cache_id = nd[sd]->cacheId;
            x.store_id = -1;
            x.reserved = 0;
        }
        s.lowQ_r = (option->alnMode == MERGE_MSA) ? false : ((s.refNum > 1) ? false : database->sequences[nd[0]->seqsIncluded[0]]->lowQuality);
        s.lowQ_q = (option->alnMode == MERGE_MSA) ? false : ((s.qryNum > 1) ? false : database->sequences[nd[1]->seqsIncluded[0]]->lowQuality);
    }
    for (int i = 0; i < n; ++i) {
        PairState &s = ps[i];
        Node *nd[2] = {nodes[i].first, nodes[i].second};
        const bool storeFreq = (s.refNum >= option->calProfileTh || s.qryNum >= option->calProfileTh) || (sides[2 * (size_t)i].cache_id >= 0 || sides[2 * (size_t)i + 1].cache_id >= 0);
        for (int sd = 0; sd < 2; ++sd) {
            twl_side &x = sides[2 * (size_t)i + sd];
            x.member_off = (int32_t)nMembers;
            if (storeFreq && x.cache_id < 0) x.store_id = nd[sd]->cacheId = ctx.nextCacheId++;
            nMembers += (size_t)x.n_members;
            stride = std::max(stride, x.len);
        }
    }
    std::vector<int32_t> members(nMembers);
    std::vector<float> weights(nMembers);
    // (levels of a few pairs stay on this thread: a parallel region of one iteration wakes every worker, and their spinning at its end
    // slowed the serial stretches that follow -- 1.3 ms against 0.3 ms for the 100 000 members of a top level)
#pragma omp parallel for schedule(dynamic, 16) if (n >= 64)
    for (int i = 0; i < n; ++i) {
        Node *nd[2] = {nodes[i].first, nodes[i].second};
        for (int sd = 0; sd < 2; ++sd) {
            const twl_side &x = sides[2 * (size_t)i + sd];
            const float groupWeight = nd[sd]->alnWeight;
            for (int m = 0; m < x.n_members; ++m) {
                const int sIdx = nd[sd]->seqsIncluded[m];
                if (sIdx < 0) { std::cerr << "ERROR: compressed sequence group in device-resident mode.\n"; exit(1); }
                members[x.member_off + m] = sIdx;
                weights[x.member_off + m] = database->sequences[sIdx]->weight / groupWeight * x.num;     // alignment-helper.cpp:27
            }
        }
    }

    // ---- device: profiles, consensus, gappy-column removal, gap penalties ----
    twl_params tp = baseParams(param);
    std::vector<int32_t> lens(2 * (size_t)n);
    const int nd = (int)ctx.stores.size();
    std::vector<std::vector<int32_t>> lensOf(nd);
    onAllStores(ctx, "twl_level_prepare", [&](int d) {
        lensOf[d].resize(2 * (size_t)n);       // every replica prepares the whole level; column info is fetched from the first only, below
        return twl_level_prepare(ctx.stores[d], &tp, option->gappyVertical, n, sides.data(), members.data(), weights.data(), stride,
                                 d == 0 ? lens.data() : lensOf[d].data(), nullptr);
    });
    for (int d = 1; d < nd; ++d)
        if (lensOf[d] != lens) { std::cerr << "ERROR: device replicas disagree on the prepared level.\n"; exit(1); }
    const bool removal = !(option->gappyVertical == 1.0);
    // The column info (consensus letter + gappy flag per original column) only matters for addGappyColumnsBack, i.e. for pairs in
    // which some column was removed: fetch it for those (at the leaf level none: single sequences have no gap columns).
    std::vector<int> needInfo;
    for (int i = 0; i < n; ++i) {
        PairState &s = ps[i];
        s.lens = {lens[2 * i], lens[2 * i + 1]};
        if (s.lens.first != s.refLen || s.lens.second != s.qryLen) needInfo.push_back(i);
    }
    // One device, one process: the paths never leave HBM -- pairs that lost no column commit their DP path as it is, the others get their
    // columns back on the device (twl_level_restore); the column info comes to the host only for a pair the device hands back (below).
    const bool procs = ctx.shard.world > 1;
    // ... and so it is with several processes when they can all-gather device blocks (Shard::exchangeDev): each rank restores its own
    // pairs on its device, the final paths travel HBM to HBM (exchangeFinalPaths below), every rank commits all of them from HBM.
    const bool devX = (nd == 1 && (ctx.shard.exchangeDev != nullptr || ctx.shard.rccl));
    const bool inPlace = (nd == 1 && ((!procs && !ctx.shard.exchange && !ctx.shard.rccl) || devX));
    auto infoOf = [&](const std::vector<int> &which) {      // consensus + removed runs of these pairs, one synchronisation
        uint8_t *info = reinterpret_cast<uint8_t *>(g_infoStage.get((size_t)2 * which.size() * stride));
        const int rc = twl_level_read_colinfo_many(firstStore(ctx), (int32_t)which.size(), which.data(), info);
        if (rc != TWL_OK) die("twl_level_read_colinfo_many", rc);
#pragma omp parallel for schedule(dynamic, 4)
        for (int t = 0; t < (int)which.size(); ++t) {
            PairState &s = ps[which[t]];
            runsAndConsensus(&info[((size_t)2 * t) * stride], s.refLen, removal, letters, s.gappy.first, s.consensus.first);
            runsAndConsensus(&info[((size_t)2 * t + 1) * stride], s.qryLen, removal, letters, s.gappy.second, s.consensus.second);
        }
    };
    if (inPlace) {}
    else if (needInfo.size() * 8 > (size_t)n) {        // most pairs: one transfer of the level's block
        uint8_t *info = reinterpret_cast<uint8_t *>(g_infoStage.get((size_t)2 * n * stride));
        const int rc = twl_level_read_colinfo(firstStore(ctx), -1, 0, info);
        if (rc != TWL_OK) die("twl_level_read_colinfo", rc);
#pragma omp parallel for schedule(dynamic, 4)
        for (int t = 0; t < (int)needInfo.size(); ++t) {
            PairState &s = ps[needInfo[t]];
            runsAndConsensus(&info[((size_t)2 * needInfo[t]) * stride], s.refLen, removal, letters, s.gappy.first, s.consensus.first);
            runsAndConsensus(&info[((size_t)2 * needInfo[t] + 1) * stride], s.qryLen, removal, letters, s.gappy.second, s.consensus.second);
        }
    } else if (!needInfo.empty()) infoOf(needInfo);      // a few pairs: their blocks only
    ctx.totals.prepare_ms += nowMs() - tPrep;

    // ---- DP with the reference's grouping and retry/defer policy (alignment-cpu.cpp:88-130) ----
    std::vector<alnPath> paths(n);
    std::vector<int16_t> errs(n, 0);
    std::vector<uint8_t> maskPlain(n, 0), maskZero(n, 0);
    int nPlain = 0, nZero = 0;
    for (int i = 0; i < n; ++i) {
        if (ps[i].refLen == 0) paths[i].assign(ps[i].qryLen, 1);
        if (ps[i].qryLen == 0) paths[i].insert(paths[i].end(), ps[i].refLen, 2);
        if (!paths[i].empty() || ps[i].lowQ_r || ps[i].lowQ_q) continue;
        const bool zg = (task == 1 || ps[i].refNum > 10000 || ps[i].qryNum > 10000);      // alignment-cpu.cpp:88
        if (zg) { maskZero[i] = 1; ++nZero; } else { maskPlain[i] = 1; ++nPlain; }
    }
    // deal the pairs, longest first (cost ~ R + Q after gappy-column removal), to the device replicas of this process or, with several
    // processes (one device each), to the ranks; everybody aligns its share
    std::vector<char> takesPart(n, 0);
    std::vector<long long> cost(n, 0);
    for (int i = 0; i < n; ++i) { takesPart[i] = (maskPlain[i] || maskZero[i]) ? 1 : 0; cost[i] = (long long)ps[i].lens.first + ps[i].lens.second; }
    const std::vector<int> owner = dealPairs(cost, takesPart, procs ? ctx.shard.world : nd);
    const int meBase = procs ? ctx.shard.rank : 0;       // replica d of this process aligns the pairs of owner meBase + d
    twl_params tz = tp;
    tz.gap_char = 0;
    if (g_alnStage.size() < (size_t)nd) g_alnStage.resize((size_t)nd);
    std::vector<double> callMs(nd, 0), kernMs(nd, 0), totMs(nd, 0);
    std::vector<uint64_t> cellsOf(nd, 0), redoOf(nd, 0);
    std::vector<char> needsHost(n, 0);
    for (int i : needInfo) needsHost[i] = 1;
    std::vector<int> handedBack;             // in-place mode: pairs the device handed back to the host (twl_level_restore returned -1)
    std::vector<uint8_t> fromDp(n, 0);       // 1: the DP path is the final path; 2: twl_level_restore made the final path, in HBM
    std::vector<int32_t> dpLen(n, 0);
    int pathStride = 1;                      // row pitch of the final paths: refLen + qryLen before removal bounds every path
    for (int i = 0; i < n; ++i) pathStride = std::max(pathStride, ps[i].refLen + ps[i].qryLen);
    onAllStores(ctx, "twl_level_align", [&](int d) {
        int8_t *aln = inPlace ? nullptr : reinterpret_cast<int8_t *>(g_alnStage[d].get((size_t)n * 2 * stride));
        std::vector<int32_t> alnLen(n);
        std::vector<int16_t> err(n);
        {
            // ONE launch for the pairs of both gap-character kinds (alignment-cpu.cpp:88 decides per pair; until round 4 one call per kind, which
            // doubled the latency of the top levels of a 100 000-leaf tree, where a few pairs of each kind meet)
            std::vector<uint8_t> mask(n, 0);
            int cnt = 0, cntZero = 0;
            for (int i = 0; i < n; ++i) if ((maskPlain[i] || maskZero[i]) && owner[i] == meBase + d) { mask[i] = 1; ++cnt; cntZero += maskZero[i]; }
            if (!cnt) return (int)TWL_OK;
            const double tCall = nowMs();
            const int r = (cntZero == cnt) ? twl_level_align(ctx.stores[d], &tz, mask.data(), aln, alnLen.data(), err.data())
                                           : twl_level_align_mixed(ctx.stores[d], &tp, mask.data(), cntZero ? maskZero.data() : nullptr, aln, alnLen.data(), err.data());
            if (r != TWL_OK) return r;
            callMs[d] += nowMs() - tCall;
            auto addStats = [&]() {
                twl_stats st{};
                if (nd == 1 || ctx.storeDev[d] != ctx.storeDev[(d + 1) % nd]) {      // per-device counters (virtual replicas share one device: see below)
                    if (twl_get_stats(ctx.storeDev[d], &st) == TWL_OK) { cellsOf[d] += st.band_cells; redoOf[d] += (uint64_t)st.n_relaunched; kernMs[d] += st.kernel_ms; totMs[d] += st.total_ms; if (d == 0 && rec.matrix_mode < 0) { rec.matrix_mode = st.matrix_mode; rec.speculative = st.speculative; memcpy(rec.kernel, st.kernel, sizeof rec.kernel); } if (d == 0) { rec.mt_predicted += st.mt_tiles_predicted; rec.mt_inline += st.mt_tiles_inline; } }
                }
            };
            addStats();
            if (d == 0) { twl_stats st0{}; if (twl_get_stats(ctx.storeDev[0], &st0) == TWL_OK) ctx.totals.nominal_cells += st0.nominal_cells; }      // (R*Q of the level's pairs, once: the classic GCUPS figure of bench.py)
            // alignment-cpu.cpp:116-129: in the deferred pass a failed pair is retried, by whoever owns it, with a larger X-drop / band limit until it passes
            if (task == 1) {
                // a retry is a call of the whole level with a one-pair mask, and a DP call zero-fills the outputs of ALL the level's pairs: with more
                // than one pair it would wipe what the others have just produced.  The deferred pass aligns one profile per level
                // (progressive.cpp:283-291); a task-1 level of several pairs is refused up front -- whether or not a pair fails -- rather than silently
                // wrong or dependent on the data (the limit is part of the contract: include/twl_msa.h)
                if (n != 1) { std::cerr << "ERROR: a deferred-pass (task 1) level must hold exactly one pair; this one has " << n << ".\n"; return (int)TWL_ERR_UNSUPPORTED; }
                for (int i = 0; i < n; ++i) {
                    if (!mask[i] || err[i] == 0) continue;
                    twl_params tr = maskZero[i] ? tz : tp;
                    const int minLen = std::min(ps[i].lens.first, ps[i].lens.second);
                    std::vector<uint8_t> one(n, 0);
                    one[i] = 1;
                    std::vector<int32_t> len1(n);
                    std::vector<int16_t> err1(n);
                    while (err[i] != 0) {
                        if (err[i] == 3) { std::cout << "There might be some bugs in the code!\n"; exit(1); }
                        if (err[i] == 2) tr.flen = std::min(static_cast<int32_t>(tr.flen * 1.2) << 1, minLen);
                        else { tr.xdrop = static_cast<int32_t>(tr.xdrop * 2); tr.flen = std::min(static_cast<int32_t>(tr.xdrop * 4) << 1, minLen); }
                        if (option->printDetail) std::cout << "Retry pair No. " << i << "\txdrop " << tr.xdrop << " flen " << tr.flen << '\n';
                        const double tRetry = nowMs();
                        const int rr = twl_level_align(ctx.stores[d], &tr, one.data(), aln, len1.data(), err1.data());
                        if (rr != TWL_OK) return rr;
                        callMs[d] += nowMs() - tRetry;
                        addStats();
                        err[i] = err1[i]; alnLen[i] = len1[i];
                    }
                }
            }
            std::vector<int32_t> fetch, fetchLen, restore, restoreLen;
            for (int i = 0; i < n; ++i) {
                if (!mask[i]) continue;
                errs[i] = err[i];
                const int32_t len = (err[i] == 0) ? alnLen[i] : 0;
                if (!inPlace) paths[i].assign(&aln[(size_t)i * 2 * stride], &aln[(size_t)i * 2 * stride] + len);
                else if (needsHost[i]) {
                    const bool lowQ = task == 0 && (ps[i].refNum == 1 || ps[i].qryNum == 1) && (ps[i].lowQ_r || ps[i].lowQ_q);      // (deferred below: its path is dropped)
                    if (len > 0 && !lowQ) { restore.push_back(i); restoreLen.push_back(len); } else paths[i].clear();
                }
                else if (len > 0) { fromDp[i] = 1; dpLen[i] = len; }
            }
            if (!restore.empty()) {             // gappy columns back on the device: these paths never leave HBM either
                std::vector<int32_t> fin(restore.size(), -1);
                const int r3 = twl_level_restore(ctx.stores[d], &tp, (int32_t)restore.size(), restore.data(), pathStride, fin.data());
                if (r3 != TWL_OK) return r3;
                for (size_t t = 0; t < restore.size(); ++t) {
                    if (fin[t] > 0) { fromDp[restore[t]] = 2; dpLen[restore[t]] = fin[t]; }
                    else { fetch.push_back(restore[t]); fetchLen.push_back(restoreLen[t]); }      // a two-sided run too large for the device: on the host, as before
                }
            }
            if (!fetch.empty()) {               // the pairs the host has to edit: their column info and paths only
                infoOf(std::vector<int>(fetch.begin(), fetch.end()));
                handedBack.insert(handedBack.end(), fetch.begin(), fetch.end());
                int8_t *blk = reinterpret_cast<int8_t *>(g_alnStage[d].get(fetch.size() * (size_t)2 * stride));
                const int r2 = twl_level_read_paths(ctx.stores[d], (int32_t)fetch.size(), fetch.data(), fetchLen.data(), blk, 2 * stride);
                if (r2 != TWL_OK) return r2;
                for (size_t t = 0; t < fetch.size(); ++t) paths[fetch[t]].assign(&blk[t * (size_t)2 * stride], &blk[t * (size_t)2 * stride] + fetchLen[t]);
            }
        }
        return (int)TWL_OK;
    });
    (void)nPlain; (void)nZero;
#ifdef TWL_DEV      // development builds: one line per pair for tools/sim_schedule.py
    if (const char *dump = getenv("TWL_DUMP_SCHEDULE")) {
        if (FILE *f = fopen(dump, "a")) {
            std::vector<uint64_t> pc(n, 0);
            if (nd == 1) (void)twl_get_pair_cells(ctx.storeDev[0], pc.data(), n);
            for (int i = 0; i < n; ++i)
                fprintf(f, "%d %d %d %d %d %d %d %llu\n", (int)ctx.levels.size(), i, nodes[i].first->seqsIncluded.empty() ? -1 : nodes[i].first->seqsIncluded[0],
                        nodes[i].second->seqsIncluded.empty() ? -1 : nodes[i].second->seqsIncluded[0], ps[i].lens.first, ps[i].lens.second, (int)errs[i], (unsigned long long)pc[i]);
            fclose(f);
        }
    }
#endif
    ctx.totals.call_ms += *std::max_element(callMs.begin(), callMs.end());           // the replicas run concurrently
    ctx.totals.total_ms += *std::max_element(totMs.begin(), totMs.end());
    rec.kernel_ms = *std::max_element(kernMs.begin(), kernMs.end());
    for (uint64_t c : cellsOf) rec.band_cells += c;
    for (uint64_t c : redoOf) rec.relaunched += c;
    // gappy columns back on the host for one pair (alignment-helper.cpp:324-375)
    auto restoreOnHost = [&](int i, alnPath &full) {
        PairState &s = ps[i];
        int alnRef = 0, alnQry = 0;
        for (auto a : paths[i]) { if (a != 1) ++alnRef; if (a != 2) ++alnQry; }
        alignment_helper::addGappyColumnsBack(paths[i], full, s.gappy, param, {alnRef, alnQry}, s.consensus);
        alnRef = alnQry = 0;
        for (auto a : full) { if (a != 1) ++alnRef; if (a != 2) ++alnQry; }
        if (alnRef != s.refLen) std::cout << "R: Post " << nodes[i].first->identifier << "(" << alnRef << "/" << s.refLen << ")\n";
        if (alnQry != s.qryLen) std::cout << "Q: Post " << nodes[i].second->identifier << "(" << alnQry << "/" << s.qryLen << ")\n";
        if ((int)full.size() > pathStride) { std::cerr << "ERROR: path longer than both profiles together.\n"; exit(1); }
    };
    for (int i : handedBack) {                // (rare: a two-sided run too large for the device) the host's result joins the others in HBM
        if (errs[i] != 0 || paths[i].empty()) continue;
        alnPath full;
        restoreOnHost(i, full);
        const int rc = twl_level_write_final(firstStore(ctx), i, full.data(), (int32_t)full.size());
        if (rc != TWL_OK) die("twl_level_write_final", rc);
        fromDp[i] = 2; dpLen[i] = (int32_t)full.size();
        paths[i].clear();
    }
    if (devX) {                               // HBM to HBM
        std::vector<int32_t> bound(n);
        for (int i = 0; i < n; ++i) bound[i] = ps[i].refLen + ps[i].qryLen;
        exchangeFinalPaths(ctx, firstStore(ctx), ctx.storeDev[0], tp, owner, takesPart, bound, pathStride, fromDp, dpLen, errs, rec);
    }
    else exchangePaths(ctx, owner, takesPart, 2 * stride, paths, errs, rec);             // several processes: everybody gets every path
    std::vector<int> fallbackPairs;
    for (int i = 0; i < n; ++i) {
        if (errs[i] == 0) continue;
        if (errs[i] == 3 || task != 0) { std::cout << "There might be some bugs in the code!\n"; exit(1); }      // (the deferred pass retried until errorType 0)
        paths[i].clear();                       // currentTask == 0: a failed pair is deferred (alignment-cpu.cpp:108-115)
        fromDp[i] = 0;
        fallbackPairs.push_back(i);
    }

    // ---- gappy columns back, then the write-back on the device ----
    const double tFin = nowMs();
    int8_t *finalPaths = reinterpret_cast<int8_t *>(g_finalStage.get((size_t)n * pathStride));
    std::vector<int32_t> finalLen(n, 0);
    std::vector<char> deferred(n, 0);
    std::vector<int> onHost;                 // pairs whose final path the host still has to make (none in the in-place mode: their paths are in HBM)
    for (int i = 0; i < n; ++i) {
        PairState &s = ps[i];
        deferred[i] = (task == 0 && (s.refNum == 1 || s.qryNum == 1) && (s.lowQ_r || s.lowQ_q)) ? 1 : 0;          // :136-144
        if (deferred[i]) { paths[i].clear(); fromDp[i] = 0; }
        if (fromDp[i]) { finalLen[i] = dpLen[i]; continue; }      // no column was removed: the DP path is the final path, and it is in HBM
        if (!paths[i].empty()) onHost.push_back(i);
    }
#pragma omp parallel for schedule(dynamic, 1) if (onHost.size() > 1)
    for (int t = 0; t < (int)onHost.size(); ++t) {
        const int i = onHost[t];
        PairState &s = ps[i];
        if (s.gappy.first.empty() && s.gappy.second.empty()) {      // nothing was removed: the DP path is the final path (addGappyColumnsBack would copy it)
            if ((int)paths[i].size() > pathStride) { std::cerr << "ERROR: path longer than both profiles together.\n"; exit(1); }
            std::copy(paths[i].begin(), paths[i].end(), &finalPaths[(size_t)i * pathStride]);
            finalLen[i] = (int32_t)paths[i].size();
            continue;
        }
        alnPath full;
        restoreOnHost(i, full);
        std::copy(full.begin(), full.end(), &finalPaths[(size_t)i * pathStride]);
        finalLen[i] = (int32_t)full.size();
    }
    const double tGappy = nowMs() - tFin;
    onAllStores(ctx, "twl_level_commit", [&](int d) { return twl_level_commit_from_dp(ctx.stores[d], finalPaths, finalLen.data(), pathStride, inPlace ? fromDp.data() : nullptr); });
#pragma omp parallel for schedule(static) if (n >= 512)
    for (int i = 0; i < n; ++i) {               // Node bookkeeping of updateFrequency / updateAlignment (alignment-helper.cpp:474-478,536-538); the pairs of a level share no node
        if (finalLen[i] == 0) continue;
        Node *a = nodes[i].first, *b = nodes[i].second;
        if (a->cacheId >= 0 && b->cacheId >= 0) b->cacheId = -1;       // merged into a's cache by the commit
        a->alnNum += b->alnNum;
        a->alnLen = finalLen[i];
        a->alnWeight += b->alnWeight;
        a->seqsIncluded.insert(a->seqsIncluded.end(), b->seqsIncluded.begin(), b->seqsIncluded.end());
        b->seqsIncluded.clear();
    }
    ctx.totals.finish_ms += nowMs() - tFin;
    double devPrep = 0, devCommit = 0;
    if (option->printDetail) twl_level_timing(firstStore(ctx), &devPrep, &devCommit);      // (waits for the write-back kernels, which the next level would otherwise overlap with)
    ctx.totals.dev_prepare_ms += devPrep;
    ctx.totals.dev_commit_ms += devCommit;
    for (int i = 0; i < n; ++i)
        if (deferred[i]) fallbackPairs.push_back(i);
    if (!fallbackPairs.empty()) alignment_helper::fallback2cpu(fallbackPairs, nodes, database, option);
    rec.level_ms = nowMs() - tPrep;
    ctx.totals.pairs += (uint64_t)n; ctx.totals.band_cells += rec.band_cells; ctx.totals.relaunched += rec.relaunched; ctx.totals.kernel_ms += rec.kernel_ms; ctx.totals.exchange_ms += rec.exchange_ms;
    ctx.levels.push_back(rec);
    if (option->printDetail)
        std::cerr << "  phases (ms): prepare " << ctx.totals.prepare_ms - before.prepare_ms << " (device " << devPrep << ") call " << ctx.totals.call_ms - before.call_ms
                  << " (kernel " << rec.kernel_ms << ", exchange " << rec.exchange_ms << ") finish " << ctx.totals.finish_ms - before.finish_ms << " (device " << devCommit
                  << ") whole " << nowMs() - tPrep << "; relaunched pairs " << ctx.totals.relaunched - before.relaunched << "; pairs with removed columns " << needInfo.size() << "; restored on the host " << handedBack.size() << "; gappy columns back " << tGappy << " ms\n";
}

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

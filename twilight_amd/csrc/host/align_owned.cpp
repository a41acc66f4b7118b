// twilight_amd/csrc/host/align_owned.cpp -- subtree ownership for a sharded run (one process per GPU).
//
// Until round 4 every rank prepared and committed ALL pairs of every level (only the DP was dealt out), so the non-DP part of a pass
// was replicated on every GPU and one collective per level sat on the critical path from the leaves up.  The pairs below a cut of the
// guide tree form independent subtrees: whoever owns a subtree aligns ALL its pairs on its own device -- profiles, DP, gappy columns,
// write-back -- without talking to anybody, level after level, exactly as a single-GPU run would (same kernel, same call).  Where the
// subtrees meet, the ranks exchange what they made ONCE: the rows of their sequences, the cached profiles and the bookkeeping of
// their subtree roots (and of the nodes they deferred).  From there on every rank holds the state a single-GPU run has after that
// level, and the few levels above the cut run as before: pairs dealt per level, final paths all-gathered (align_gpu.cpp).
//
// The cut is a LEVEL of the reference's schedule (progressive.cpp:109-124): the highest one that still leaves 8 subtrees per rank
// (longest-processing-time deal by number of pairs), so the part above it is at most a few dozen pairs.  The reference deals batches
// of a level to devices (/root/reference/src/hip/alignment-gpu.hip.cpp:239-254); pairs of different subtrees below the cut are as
// independent as pairs of one level, and the result does not depend on who aligned what: a pair's alignment is a function of its two
// sub-alignments only.  What must be kept is the ORDER of the deferred nodes (they are sorted later with an unstable comparison-equal
// tail, progressive.cpp:276-279): every deferred node travels with (level, index of its pair in the level) and the lists are merged
// in that order.
#include "align_gpu.hpp"

#include <algorithm>
#include <cstring>
#include <iostream>
#include <numeric>
#include <unordered_map>

namespace msa {
namespace progressive {
namespace gpu {

namespace {

struct Writer {
    std::vector<char> b;
    template <class T> void put(const T &v) { const size_t at = b.size(); b.resize(at + sizeof(T)); memcpy(&b[at], &v, sizeof(T)); }
    void bytes(const void *p, size_t n) { const size_t at = b.size(); b.resize(at + n); if (n) memcpy(&b[at], p, n); }
    void align8() { b.resize((b.size() + 7) & ~(size_t)7, 0); }
};
struct Reader {
    const char *p; size_t n, at = 0;
    template <class T> T get() { T v; need(sizeof(T)); memcpy(&v, p + at, sizeof(T)); at += sizeof(T); return v; }
    const char *bytes(size_t k) { need(k); const char *r = p + at; at += k; return r; }
    void align8() { at = (at + 7) & ~(size_t)7; }
    void need(size_t k) const { if (at + k > n) { std::cerr << "ERROR: malformed subtree block from another rank.\n"; exit(1); } }
};

void die(const char *what, int rc)
{
    std::cerr << "ERROR: " << what << " failed (" << rc << "): " << twl_last_error() << '\n';
    exit(1);
}

// all-gather of byte strings of different lengths: sizes first, then the strings padded to the longest
std::vector<std::vector<char>> allGatherBlobs(RunCtx &ctx, const std::vector<char> &mine)
{
    const Shard &sh = ctx.shard;
    auto gather = [&](const void *send, int64_t bytes, void *recv) -> int {
        if (sh.rccl) return twl_comm_all_gather_host(selectedDevices()[0], send, recv, bytes);
        return sh.exchange(sh.user, send, bytes, recv);
    };
    std::vector<int64_t> sizes((size_t)sh.world, 0);
    const int64_t mySize = (int64_t)mine.size();
    int rc = gather(&mySize, (int64_t)sizeof(int64_t), sizes.data());
    if (rc != 0) { std::cerr << "ERROR: exchange of the subtree block sizes failed (" << rc << ").\n"; exit(1); }
    const size_t blk = ((size_t)*std::max_element(sizes.begin(), sizes.end()) + 255) & ~(size_t)255;
    std::vector<char> send(blk, 0), recv(blk * (size_t)sh.world);
    memcpy(send.data(), mine.data(), mine.size());
    rc = gather(send.data(), (int64_t)blk, recv.data());
    if (rc != 0) { std::cerr << "ERROR: exchange of the subtree blocks failed (" << rc << ").\n"; exit(1); }
    std::vector<std::vector<char>> out((size_t)sh.world);
    for (int r = 0; r < sh.world; ++r) out[r].assign(recv.begin() + (ptrdiff_t)(blk * (size_t)r), recv.begin() + (ptrdiff_t)(blk * (size_t)r + (size_t)sizes[r]));
    return out;
}

constexpr uint64_t kMagic = 0x54574C4F574E4544ull;      // "TWLOWNED"

}  // namespace

// Levels [0, returned) of the main pass are done when this returns: every rank aligned the subtrees it owns and imported the others'.
size_t ownedPrefix(Tree *T, std::vector<NodePairVec> &levels, SequenceDB *database, Option *option, Params &param)
{
    RunCtx &ctx = ctxOf(database);
    const Shard sh = ctx.shard;
    if (sh.world <= 1 || database->currentTask != 0 || !(sh.rccl || sh.exchange) || ctx.finished) return 0;
    if (option->testNoOwnership) return 0;         // (--test-no-ownership: the round-3 behaviour, every level dealt and exchanged)
    const OwnershipPlan plan = planOwnership(T, levels, sh.world);
    const int cut = plan.cut;
    if (cut < 0) return 0;
    std::unordered_map<Node *, int> idOf;      // a number for every node of the schedule, the same on every rank
    std::vector<Node *> nodeOf;
    for (auto &lv : levels) for (auto &pr : lv) for (Node *x : {pr.first, pr.second}) if (!idOf.count(x)) { idOf.emplace(x, (int)nodeOf.size()); nodeOf.push_back(x); }
    long long pairsBelow = 0, pairsAbove = 0;
    for (int l = 0; l < (int)levels.size(); ++l) (l <= cut ? pairsBelow : pairsAbove) += (long long)levels[l].size();
    if (option->printDetail)
        std::cerr << "Sharded run, rank " << sh.rank << " of " << sh.world << ": levels 1-" << cut + 1 << " by subtree ownership (" << plan.subtrees << " subtrees, " << plan.load[sh.rank]
                  << " of " << pairsBelow << " pairs here), " << pairsAbove << " pairs above dealt per level\n";

    // ---- phase 1: my subtrees, alone ----
    struct Deferred { int level, idx, node; };
    std::vector<Deferred> myDeferred;
    std::vector<int> touched;                               // nodes my pairs touched, in first-touch order
    std::vector<char> seen(nodeOf.size(), 0);
    const size_t recBase = ctx.levels.size();
    ctx.shard = Shard{};                                    // (the level kernel runs as on one GPU: nothing is dealt, nothing exchanged)
    const double t0 = nowMs();
    for (int l = 0; l <= cut; ++l) {
        NodePairVec mine;
        std::unordered_map<Node *, int> origIdx;
        for (int i = 0; i < (int)levels[l].size(); ++i)
            if (plan.owner[l][i] == sh.rank) { mine.push_back(levels[l][i]); origIdx[levels[l][i].second] = i; }
        const size_t fb0 = database->fallback_nodes.size();
        if (!mine.empty()) {
            for (auto &pr : mine) for (Node *x : {pr.first, pr.second}) { const int k = idOf[x]; if (!seen[k]) { seen[k] = 1; touched.push_back(k); } }
            updateNode(T, mine, database);
            alignmentKernel_Resident(T, mine, database, option, param);
        } else {
            LevelRecord rec;
            ctx.levels.push_back(rec);
        }
        for (size_t k = fb0; k < database->fallback_nodes.size(); ++k) {
            Node *b = database->fallback_nodes[k];
            myDeferred.push_back({l, origIdx.count(b) ? origIdx[b] : 0, idOf[b]});
        }
        if (option->printDetail) std::cerr << "Level " << l + 1 << ", aligned " << mine.size() << " of " << levels[l].size() << " pairs here (subtree ownership)\n";
    }
    const double tOwn = nowMs() - t0;
    ctx.shard = sh;

    // ---- the one exchange: what my subtrees have become ----
    const double tX = nowMs();
    twl_store *store = ctx.stores.empty() ? nullptr : ctx.stores[0];
    if (!store) { uploadSequences(database, option); store = ctx.stores[0]; }      // (a rank without a pair below the cut)
    const int P = (option->type == 'n') ? 6 : 22;
    Writer w;
    // fixed header: magic, cut, how the rows travel, their byte count (patched in below) -- everything a receiver must believe before it sizes a buffer
    w.put<uint64_t>(kMagic);
    w.put<int32_t>(cut); w.put<int32_t>(0 /* rows on device, patched */);
    const size_t rowBytesAt = w.b.size();
    w.put<uint64_t>(0);
    w.put<int32_t>((int32_t)touched.size());
    std::vector<int32_t> mySeqs;
    for (int k : touched) {
        Node *n = nodeOf[k];
        w.put<int32_t>(k); w.put<int32_t>(n->alnLen); w.put<int32_t>(n->alnNum); w.put<float>(n->alnWeight);
        w.put<int32_t>(n->cacheId >= 0 ? 1 : 0); w.put<int32_t>((int32_t)n->seqsIncluded.size());
        w.bytes(n->seqsIncluded.data(), n->seqsIncluded.size() * sizeof(int));
        for (int s : n->seqsIncluded) mySeqs.push_back(s);
        if (n->cacheId >= 0) {
            int32_t len = 0;
            int rc = twl_store_read_cache(store, n->cacheId, nullptr, &len);
            if (rc != TWL_OK) die("twl_store_read_cache", rc);
            std::vector<float> flat((size_t)len * P);
            if ((rc = twl_store_read_cache(store, n->cacheId, flat.data(), &len)) != TWL_OK) die("twl_store_read_cache", rc);
            w.put<int32_t>(len);
            w.bytes(flat.data(), flat.size() * sizeof(float));
        }
    }
    w.put<int32_t>((int32_t)myDeferred.size());
    for (auto &d : myDeferred) { w.put<int32_t>(d.level); w.put<int32_t>(d.idx); w.put<int32_t>(d.node); }
    for (int l = 0; l <= cut; ++l) {
        const LevelRecord &r = ctx.levels[recBase + (size_t)l];
        w.put<int32_t>(r.pairs); w.put<uint64_t>(r.band_cells); w.put<uint64_t>(r.relaunched); w.put<double>(r.kernel_ms); w.put<double>(r.level_ms);
        w.put<int32_t>(r.mt_predicted); w.put<int32_t>(r.mt_inline); w.put<int32_t>(r.matrix_mode); w.put<int32_t>(r.speculative);
        w.bytes(r.kernel, sizeof r.kernel);
    }
    // The rows travel HBM to HBM when the run has a device collective (the library's RCCL communicator, or the caller's device callback): packed into one
    // device block per rank, all-gathered, unpacked by a kernel -- 10 000 x 10 kbp at 8 ranks is ~140 MB of rows at the cut, which would otherwise cross
    // PCIe twice.  Without one (host callback only) they ride in the host block.
    const bool rowsOnDevice = sh.rccl || sh.exchangeDev != nullptr;
    std::vector<int32_t> myLens;
    size_t myRowBytes = 0;
    {   // rows (and the low-quality flags fallback2cpu may have cleared) of the sequences in my subtrees
        std::sort(mySeqs.begin(), mySeqs.end());
        mySeqs.erase(std::unique(mySeqs.begin(), mySeqs.end()), mySeqs.end());
        myLens.resize(mySeqs.size());
        int rc = twl_store_read_rows_of(store, (int32_t)mySeqs.size(), mySeqs.data(), nullptr, myLens.data());
        if (rc != TWL_OK) die("twl_store_read_rows_of", rc);
        for (int32_t x : myLens) myRowBytes += (size_t)x;
        w.put<int32_t>((int32_t)mySeqs.size());
        w.bytes(mySeqs.data(), mySeqs.size() * sizeof(int32_t));
        w.bytes(myLens.data(), myLens.size() * sizeof(int32_t));
        for (int32_t s : mySeqs) w.put<uint8_t>(database->sequences[s]->lowQuality ? 1 : 0);
        w.align8();
        { const uint64_t rb = (uint64_t)myRowBytes; memcpy(&w.b[rowBytesAt], &rb, 8); const int32_t od = rowsOnDevice ? 1 : 0; memcpy(&w.b[rowBytesAt - 4], &od, 4); }
        if (!rowsOnDevice) {
            const size_t at = w.b.size();
            w.b.resize(at + myRowBytes);
            if ((rc = twl_store_read_rows_of(store, (int32_t)mySeqs.size(), mySeqs.data(), w.b.data() + at, myLens.data())) != TWL_OK) die("twl_store_read_rows_of", rc);
        }
    }
    std::vector<std::vector<char>> blobs = allGatherBlobs(ctx, w.b);
    // ... then the device blocks: every rank knows every rank's row bytes from the host blocks
    void *rowsRecv = nullptr;
    size_t rowsBlk = 0;
    if (rowsOnDevice) {
        for (int r = 0; r < sh.world; ++r) {
            // the header is checked BEFORE its byte count sizes anything: magic, cut, and the same idea of how the rows travel
            Reader probe{blobs[r].data(), blobs[r].size()};
            if (probe.get<uint64_t>() != kMagic || probe.get<int32_t>() != cut) { std::cerr << "ERROR: subtree block of rank " << r << " does not belong to this run.\n"; exit(1); }
            if (probe.get<int32_t>() != 1) { std::cerr << "ERROR: the ranks disagree on how the subtrees' rows travel.\n"; exit(1); }
            const uint64_t rb = probe.get<uint64_t>();
            if (rb > ((uint64_t)1 << 40)) { std::cerr << "ERROR: malformed subtree block from rank " << r << ".\n"; exit(1); }
            rowsBlk = std::max(rowsBlk, (size_t)rb);
        }
        rowsBlk = (rowsBlk + 255) & ~(size_t)255;
        if (rowsBlk > 0) {
            void *rowsSend = nullptr;
            int rc = twl_store_exchange_buffers(store, (int64_t)rowsBlk, (int64_t)(rowsBlk * (size_t)sh.world), &rowsSend, &rowsRecv);
            if (rc != TWL_OK) die("twl_store_exchange_buffers", rc);
            if ((rc = twl_store_rows_to_block(store, (int32_t)mySeqs.size(), mySeqs.data(), rowsSend, myLens.data())) != TWL_OK) die("twl_store_rows_to_block", rc);
            const int xrc = sh.rccl ? twl_comm_all_gather(selectedDevices()[0], rowsSend, rowsRecv, (int64_t)rowsBlk) : sh.exchangeDev(sh.userDev, rowsSend, (int64_t)rowsBlk, rowsRecv);
            if (xrc != 0) { std::cerr << "ERROR: device exchange of the subtrees' rows failed (" << xrc << "): " << twl_last_error() << '\n'; exit(1); }
        }
    }

    // ---- import the others' subtrees; merge the deferred lists and the level records ----
    struct Tagged { int level, idx; Node *node; };
    std::vector<Tagged> deferredAll;
    std::vector<LevelRecord> merged((size_t)cut + 1);
    for (int r = 0; r < sh.world; ++r) {
        Reader rd{blobs[r].data(), blobs[r].size()};
        if (rd.get<uint64_t>() != kMagic || rd.get<int32_t>() != cut) { std::cerr << "ERROR: subtree block of rank " << r << " does not belong to this run.\n"; exit(1); }
        const bool onDev = rd.get<int32_t>() != 0;
        const uint64_t total = rd.get<uint64_t>();
        if (onDev != rowsOnDevice) { std::cerr << "ERROR: the ranks disagree on how the subtrees' rows travel.\n"; exit(1); }
        const int nTouched = rd.get<int32_t>();
        for (int t = 0; t < nTouched; ++t) {
            const int k = rd.get<int32_t>();
            if (k < 0 || k >= (int)nodeOf.size()) { std::cerr << "ERROR: malformed subtree block from rank " << r << ".\n"; exit(1); }
            Node *n = nodeOf[k];
            const int alnLen = rd.get<int32_t>(), alnNum = rd.get<int32_t>();
            const float alnWeight = rd.get<float>();
            const int hasCache = rd.get<int32_t>(), nSeq = rd.get<int32_t>();
            const char *seqs = rd.bytes((size_t)nSeq * sizeof(int));
            int32_t clen = 0;
            const char *cdata = nullptr;
            if (hasCache) { clen = rd.get<int32_t>(); cdata = rd.bytes((size_t)clen * P * sizeof(float)); }
            if (r == sh.rank) continue;
            n->alnLen = alnLen; n->alnNum = alnNum; n->alnWeight = alnWeight;
            n->seqsIncluded.resize((size_t)nSeq);
            if (nSeq) memcpy(n->seqsIncluded.data(), seqs, (size_t)nSeq * sizeof(int));
            n->cacheId = -1;
            if (hasCache) {
                std::vector<float> flat((size_t)clen * P);      // (the block holds them unaligned)
                if (!flat.empty()) memcpy(flat.data(), cdata, flat.size() * sizeof(float));
                n->cacheId = ctx.nextCacheId++;
                const int rc = twl_store_write_cache(store, n->cacheId, flat.data(), clen);
                if (rc != TWL_OK) die("twl_store_write_cache", rc);
            }
        }
        const int nDef = rd.get<int32_t>();
        for (int t = 0; t < nDef; ++t) {
            const int level = rd.get<int32_t>(), idx = rd.get<int32_t>(), k = rd.get<int32_t>();
            if (k < 0 || k >= (int)nodeOf.size()) { std::cerr << "ERROR: malformed subtree block from rank " << r << ".\n"; exit(1); }
            deferredAll.push_back({level, idx, nodeOf[k]});
        }
        for (int l = 0; l <= cut; ++l) {
            LevelRecord &m = merged[l];
            m.pairs += rd.get<int32_t>(); m.band_cells += rd.get<uint64_t>(); m.relaunched += rd.get<uint64_t>();
            m.kernel_ms = std::max(m.kernel_ms, rd.get<double>()); m.level_ms = std::max(m.level_ms, rd.get<double>());      // the ranks ran side by side
            m.mt_predicted += rd.get<int32_t>(); m.mt_inline += rd.get<int32_t>();
            const int mm = rd.get<int32_t>(), sp = rd.get<int32_t>();
            const char *kn = rd.bytes(sizeof m.kernel);
            if (m.matrix_mode < 0 && mm >= 0) { m.matrix_mode = mm; m.speculative = sp; memcpy(m.kernel, kn, sizeof m.kernel); }
        }
        const int nSeqs = rd.get<int32_t>();
        const char *ids = rd.bytes((size_t)nSeqs * sizeof(int32_t));
        const char *lens = rd.bytes((size_t)nSeqs * sizeof(int32_t));
        const char *lowq = rd.bytes((size_t)nSeqs);
        rd.align8();
        const char *rows = onDev ? nullptr : rd.bytes((size_t)total);
        if (r == sh.rank || nSeqs == 0) continue;
        std::vector<int32_t> idv((size_t)nSeqs), lenv((size_t)nSeqs);
        memcpy(idv.data(), ids, idv.size() * sizeof(int32_t));
        memcpy(lenv.data(), lens, lenv.size() * sizeof(int32_t));
        for (int t = 0; t < nSeqs; ++t) {
            if (idv[t] < 0 || idv[t] >= (int)database->sequences.size()) { std::cerr << "ERROR: malformed subtree block from rank " << r << ".\n"; exit(1); }
            database->sequences[idv[t]]->lowQuality = lowq[t] != 0;
        }
        const int rc = onDev ? twl_store_rows_from_block(store, nSeqs, idv.data(), lenv.data(), static_cast<const char *>(rowsRecv) + rowsBlk * (size_t)r)
                             : twl_store_write_rows(store, nSeqs, idv.data(), rows, lenv.data());
        if (rc != TWL_OK) die(onDev ? "twl_store_rows_from_block" : "twl_store_write_rows", rc);
    }
    std::stable_sort(deferredAll.begin(), deferredAll.end(), [](const Tagged &a, const Tagged &b) { return a.level != b.level ? a.level < b.level : a.idx < b.idx; });
    database->fallback_nodes.clear();
    for (auto &d : deferredAll) database->fallback_nodes.push_back(d.node);
    const double xMs = nowMs() - tX;
    // the records and totals of the owned levels as one rank would have them
    uint64_t myCells = 0, myPairs = 0, myRedo = 0;
    double myKernel = 0;
    for (int l = 0; l <= cut; ++l) { const LevelRecord &r = ctx.levels[recBase + (size_t)l]; myCells += r.band_cells; myPairs += (uint64_t)r.pairs; myRedo += r.relaunched; myKernel += r.kernel_ms; }
    ctx.totals.band_cells -= myCells; ctx.totals.pairs -= myPairs; ctx.totals.relaunched -= myRedo; ctx.totals.kernel_ms -= myKernel;
    merged[cut].exchange_ms += xMs;
    for (int l = 0; l <= cut; ++l) {
        ctx.levels[recBase + (size_t)l] = merged[l];
        ctx.totals.band_cells += merged[l].band_cells; ctx.totals.pairs += (uint64_t)merged[l].pairs; ctx.totals.relaunched += merged[l].relaunched; ctx.totals.kernel_ms += merged[l].kernel_ms;
    }
    ctx.totals.exchange_ms += xMs;
    if (option->printDetail)
        std::cerr << "Subtrees of levels 1-" << cut + 1 << " aligned in " << tOwn << " ms here, exchanged in " << xMs << " ms (" << w.b.size() << " bytes from this rank); "
                  << database->fallback_nodes.size() << " deferred nodes\n";
    return (size_t)cut + 1;
}

}  // namespace gpu
}  // namespace progressive
}  // namespace msa

// twilight_amd/csrc/host/helpers.cpp -- per-pair pre/post-processing around the DP.
// Behavioural mirror of /root/reference/src/alignment-helper.cpp:8-591 and of the non-DP parts of
// alignment-cpu.cpp:50-93,136-175.  Float/double mixing follows the reference expression by expression because
// column counts, gappy-column decisions and gap penalties feed the bit-exact DP.
#include "twl_host.hpp"
#include <cstring>
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <iostream>
#include <mutex>

namespace msa {
namespace alignment_helper {


static std::mutex g_mapMutex;      // plays database->mapMutex (alignment-helper.cpp:406,420)

// alignment-helper.cpp:8-72
static void profileOfSide(float *dst, Node *node, SequenceDB *db, Option *option, int P, int len, int num, bool storeFreq)
{
    const float groupWeight = node->alnWeight;
    if (!node->msaFreq.empty()) {
        for (int t = 0; t < len; ++t)
            for (int v = 0; v < P; ++v) dst[P * t + v] = node->msaFreq[t][v] / groupWeight * num;
        return;
    }
    // Same per-column summation order as the reference (member by member, :23-34); columns are split into chunks so that a
    // single wide pair (upper tree levels) still uses every core and each member row is read with unit stride.
    const int nSeq = (int)node->seqsIncluded.size();
    std::vector<float> wOf(nSeq);
    std::vector<const char *> rowOf(nSeq);
    for (int s = 0; s < nSeq; ++s) {
        auto *q = db->sequences[node->seqsIncluded[s]];
        wOf[s] = q->weight / groupWeight * num;
        rowOf[s] = q->alnStorage[q->storage];
    }
    const int chunk = 512, nChunks = (len + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) if (nChunks > 3 && nSeq > 1)
    for (int c = 0; c < nChunks; ++c) {
        const int t0 = c * chunk, t1 = std::min(len, t0 + chunk);
        for (int s = 0; s < nSeq; ++s) {
            const float w = wOf[s];
            const char *row = rowOf[s];
            for (int t = t0; t < t1; ++t) {
                const int li = letterIdx(option->type, (char)toupper((unsigned char)row[t]));
                dst[P * t + li] += 1.0 * w;
            }
        }
    }
    if (storeFreq) {
        node->msaFreq.assign(len, std::vector<float>(P, 0.0f));
        for (int t = 0; t < len; ++t)
            for (int v = 0; v < P; ++v) node->msaFreq[t][v] = dst[P * t + v] / num * groupWeight;
    }
}

void calculateProfile(float *profile, NodePair &nodes, SequenceDB *database, Option *option, int32_t memLen)
{
    const int P = (option->type == 'n') ? 6 : 22;
    const int refNum = nodes.first->getAlnNum(database->currentTask), qryNum = nodes.second->getAlnNum(database->currentTask);
    const int refLen = nodes.first->getAlnLen(database->currentTask), qryLen = nodes.second->getAlnLen(database->currentTask);
    const bool storeFreq = (refNum >= option->calProfileTh || qryNum >= option->calProfileTh) || (!nodes.first->msaFreq.empty() || !nodes.second->msaFreq.empty());
    profileOfSide(profile, nodes.first, database, option, P, refLen, refNum, storeFreq);
    profileOfSide(profile + (size_t)P * memLen, nodes.second, database, option, P, qryLen, qryNum, storeFreq);
}

// alignment-helper.cpp:74-166
static void gappyRuns(const float *side, int P, int len, int num, float thr, IntPairVec &runs)
{
    int start = -1, length = 0;
    for (int i = 0; i < len; ++i) {
        if (side[P * i + P - 1] / num > thr) {
            if (start == -1) { start = i; length = 1; }
            else ++length;
        } else if (start != -1) {
            runs.push_back({start, length});
            start = -1;
            length = 0;
        }
    }
    if (start != -1) runs.push_back({start, length});
}

static int compactSide(float *side, int P, int orgLen, const IntPairVec &runs)
{
    int org = 0, dst = 0;
    size_t g = 0;
    while (org < orgLen) {
        if (g < runs.size() && org == runs[g].first) { org += runs[g].second; ++g; }
        else {
            for (int t = 0; t < P; ++t) side[P * dst + t] = side[P * org + t];
            ++dst;
            ++org;
        }
    }
    for (int z = dst; z < orgLen; ++z)
        for (int t = 0; t < P; ++t) side[P * z + t] = 0;
    return dst;
}

void removeGappyColumns(float *hostFreq, NodePair &nodes, Option *option, std::pair<IntPairVec, IntPairVec> &gappyColumns, int32_t memLen,
                        IntPair &lens, int currentTask)
{
    const float thr = option->gappyVertical;
    if (thr == 1.0) return;
    const int P = (option->type == 'n') ? 6 : 22;
    const int refNum = nodes.first->getAlnNum(currentTask), qryNum = nodes.second->getAlnNum(currentTask);
    float *refSide = hostFreq, *qrySide = hostFreq + (size_t)P * memLen;
    gappyRuns(refSide, P, lens.first, refNum, thr, gappyColumns.first);
    gappyRuns(qrySide, P, lens.second, qryNum, thr, gappyColumns.second);
    if (!gappyColumns.first.empty()) lens.first = compactSide(refSide, P, lens.first, gappyColumns.first);
    if (!gappyColumns.second.empty()) lens.second = compactSide(qrySide, P, lens.second, gappyColumns.second);
}

// alignment-helper.cpp:168-219 (ClustalW-style position-specific gap penalties)
void calculatePSGP(float *hostFreq, float *hostGapOp, float *hostGapEx, NodePair &nodes, SequenceDB *database, Option *option, int memLen,
                   IntPair offset, IntPair lens, Params &param)
{
    const int32_t refLen = lens.first, qryLen = lens.second;
    const int32_t offsetf = offset.first, offsetg = offset.second;
    const int32_t refNum = nodes.first->getAlnNum(database->currentTask);
    const int32_t qryNum = nodes.second->getAlnNum(database->currentTask);
    const int32_t P = (option->type == 'n') ? 6 : 22;
    const float scale = (option->type == 'n') ? 0.5 : 1.0;
    const float min_gapExtend = param.gapExtend * 0.2;
    const float min_gapOpen = param.gapOpen * 0.1;
#pragma omp parallel for schedule(static) if (memLen > 8192)
    for (int s = 0; s < memLen; ++s) {
        if (s < refLen) {
            const float g = hostFreq[offsetf + P * s + P - 1];
            if (g > 0) {
                hostGapOp[offsetg + s] = std::min(min_gapOpen, static_cast<float>(param.gapOpen * scale * ((refNum - g) * 1.0 / refNum)));
                hostGapEx[offsetg + s] = std::min(min_gapExtend, static_cast<float>(param.gapExtend * ((refNum - g) * 1.0 / refNum)));
            } else {
                hostGapOp[offsetg + s] = param.gapOpen;
                hostGapEx[offsetg + s] = param.gapExtend;
            }
        } else {
            hostGapOp[offsetg + s] = 0.0;
            hostGapEx[offsetg + s] = 0.0;
        }
        if (s < qryLen) {
            const float g = hostFreq[offsetf + P * (memLen + s) + P - 1];
            if (g > 0) {
                hostGapOp[offsetg + memLen + s] = std::min(min_gapOpen, static_cast<float>(param.gapOpen * scale * ((qryNum - g) * 1.0 / qryNum)));
                hostGapEx[offsetg + memLen + s] = std::min(min_gapExtend, static_cast<float>(param.gapExtend * ((qryNum - g) * 1.0 / qryNum)));
            } else {
                hostGapOp[offsetg + memLen + s] = param.gapOpen;
                hostGapEx[offsetg + memLen + s] = param.gapExtend;
            }
        } else {
            hostGapOp[offsetg + memLen + s] = 0.0;
            hostGapEx[offsetg + memLen + s] = 0.0;
        }
    }
}

// alignment-helper.cpp:221-241: first strict maximum over the letters; all-zero column -> N / X
void getConsensus(Option *option, float *profile, std::string &consensus, int len)
{
    static const char bases[5] = {'A', 'C', 'G', 'T', 'N'};
    static const char acids[21] = {'A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M', 'N', 'P', 'Q', 'R', 'S', 'T', 'V', 'W', 'Y', 'X'};
    const char *lut = (option->type == 'n') ? bases : acids;
    const int P = (option->type == 'n') ? 6 : 22;
    consensus.reserve(len);
    for (int i = 0; i < len; ++i) {
        int best = P - 2;
        float bestCount = 0;
        for (int j = 0; j < P - 2; ++j)
            if (profile[P * i + j] > bestCount) { bestCount = profile[P * i + j]; best = j; }
        consensus.push_back(lut[best]);
    }
}

// alignment-helper.cpp:243-322: small affine NW over two consensus substrings, free leading gaps, traceback prefers M, then Y(1), then X(2).
// Raw form: the two substrings by pointer, the path (forward order) into `out` (room for m + n codes), matrices in the caller's scratch
// (addGappyColumnsBack calls this thousands of times per pair at the top of the tree: no allocation per call).
struct PairwiseScratch { std::vector<float> M, X, Y; std::vector<int8_t> tb; std::vector<int> idx2; };
static int pairwiseGlobalRaw(const char *seq1, int m, const char *seq2, int n, int8_t *out, Params &param, PairwiseScratch &sc)
{
    const char type = (param.matrixSize == 5) ? 'n' : 'p';
    const float gap_open = param.gapOpen, gap_extend = param.gapExtend;
    const size_t W = (size_t)n + 1, cellsN = ((size_t)m + 1) * W;
    if (sc.M.size() < cellsN) { sc.M.resize(cellsN); sc.X.resize(cellsN); sc.Y.resize(cellsN); sc.tb.resize(cellsN); }
    if (sc.idx2.size() < (size_t)n + 1) sc.idx2.resize((size_t)n + 1);
    float *M = sc.M.data(), *X = sc.X.data(), *Y = sc.Y.data();
    int8_t *tb = sc.tb.data();
    M[0] = X[0] = Y[0] = 0.0f; tb[0] = 0;
    for (int i = 1; i <= m; ++i) { M[i * W] = 0; X[i * W] = M[i * W]; Y[i * W] = -1e9; tb[i * W] = 2; }
    for (int j = 1; j <= n; ++j) { M[j] = 0; Y[j] = M[j]; X[j] = -1e9; tb[j] = 1; }
    int *idx2 = sc.idx2.data();      // (letter indices once per letter, not once per cell)
    for (int j = 0; j < n; ++j) idx2[j] = letterIdx(type, (char)toupper((unsigned char)seq2[j]));
    for (int i = 1; i <= m; ++i) {
        const int a = letterIdx(type, (char)toupper((unsigned char)seq1[i - 1]));
        const float *row = param.scoringMatrix[a];
        for (int j = 1; j <= n; ++j) {
            const float base = row[idx2[j - 1]];
            const size_t c = i * W + j, up = (i - 1) * W + j, left = i * W + j - 1, diag = (i - 1) * W + j - 1;
            M[c] = base + std::max({M[diag], X[diag], Y[diag]});
            X[c] = std::max(M[up] + gap_open, X[up] + gap_extend);
            Y[c] = std::max(M[left] + gap_open, Y[left] + gap_extend);
            const float best = std::max({M[c], X[c], Y[c]});
            tb[c] = (best == M[c]) ? 0 : ((best == Y[c]) ? 1 : 2);
        }
    }
    int len = 0;
    for (int i = m, j = n; i > 0 || j > 0;) {
        const int8_t d = tb[i * W + j];
        out[len++] = d;
        if (d == 0) { --i; --j; }
        else if (d == 1) --j;
        else --i;
    }
    std::reverse(out, out + len);
    return len;
}

void pairwiseGlobal(const std::string &seq1, const std::string &seq2, alnPath &path, Params &param)
{
    PairwiseScratch sc;
    path.resize(seq1.size() + seq2.size());
    const int len = pairwiseGlobalRaw(seq1.data(), (int)seq1.size(), seq2.data(), (int)seq2.size(), path.data(), param, sc);
    path.resize((size_t)len);
}

// alignment-helper.cpp:324-375.  The reference walks the path once, keeping the ORIGINAL column index of either side, and inserts a removed
// run when that index reaches the run's start (both sides at the same step: the two runs are aligned to each other by pairwiseGlobal).
// The index of a side at step a is (columns the path has consumed before a) + (lengths of the runs inserted so far), so run g is
// inserted at the first step whose consumed-column count equals start_g - (lengths of the runs before g): every run finds its step by
// a binary search in the prefix counts, independently, and the output is the path with the run segments spliced in -- same result,
// block copies instead of one push_back per column (paths reach 10^5 columns at the top of the tree).
void addGappyColumnsBack(alnPath &before, alnPath &after, std::pair<IntPairVec, IntPairVec> &gappy, Params &param, IntPair, stringPair orgSeqs)
{
    const size_t n = before.size();
    std::vector<int32_t> cR(n + 1), cQ(n + 1);
    cR[0] = cQ[0] = 0;
    for (size_t a = 0; a < n; ++a) {
        const int8_t c = before[a];
        if (c < 0 || c > 2) std::cerr << "ERROR: Undefined TB Path: " << (int)c << '\n';
        cR[a + 1] = cR[a] + ((c == 0 || c == 2) ? 1 : 0);
        cQ[a + 1] = cQ[a] + ((c == 0 || c == 1) ? 1 : 0);
    }
    // the step of every run (runs a side never reaches are not inserted, as in the walk)
    auto stepsOf = [&](const IntPairVec &runs, const std::vector<int32_t> &cnt) {
        std::vector<size_t> at;
        long long before_len = 0;
        for (const auto &run : runs) {
            const long long p = (long long)run.first - before_len;
            auto it = std::lower_bound(cnt.begin(), cnt.end(), (int32_t)std::max<long long>(p, -1));
            if (p < 0 || it == cnt.end() || *it != p) break;
            at.push_back((size_t)(it - cnt.begin()));
            before_len += run.second;
        }
        return at;
    };
    const std::vector<size_t> aR = stepsOf(gappy.first, cR), aQ = stepsOf(gappy.second, cQ);
    struct Event { size_t a; int r, q; };
    std::vector<Event> ev;
    for (size_t i = 0, j = 0; i < aR.size() || j < aQ.size();) {
        if (j >= aQ.size() || (i < aR.size() && aR[i] < aQ[j])) { ev.push_back({aR[i], (int)i, -1}); ++i; }
        else if (i >= aR.size() || aQ[j] < aR[i]) { ev.push_back({aQ[j], -1, (int)j}); ++j; }
        else { ev.push_back({aR[i], (int)i, (int)j}); ++i; ++j; }
    }
    // the segment of every event; runs of both sides at one step are aligned to each other (thousands of small alignments per pair at the
    // top of the tree: their paths go into one arena, a contiguous share of them per thread, scratch matrices per thread)
    std::vector<size_t> segLen(ev.size()), arenaOff(ev.size(), 0);
    std::vector<int> bothIdx;
    size_t arenaBytes = 0;
    for (size_t e = 0; e < ev.size(); ++e) {
        if (ev[e].r >= 0 && ev[e].q >= 0) {
            bothIdx.push_back((int)e);
            arenaOff[e] = arenaBytes;
            arenaBytes += (size_t)gappy.first[ev[e].r].second + (size_t)gappy.second[ev[e].q].second;
        } else segLen[e] = (size_t)(ev[e].r >= 0 ? gappy.first[ev[e].r].second : gappy.second[ev[e].q].second);
    }
    std::vector<int8_t> arena(arenaBytes);
    const bool nested = omp_in_parallel();
    const int nBoth = (int)bothIdx.size();
#pragma omp parallel if (!nested && nBoth > 20000)
    {
        PairwiseScratch sc;
#pragma omp for schedule(static)
        for (int b = 0; b < nBoth; ++b) {
            const size_t e = (size_t)bothIdx[b];
            const IntPair &rr = gappy.first[ev[e].r], &qq = gappy.second[ev[e].q];
            segLen[e] = (size_t)pairwiseGlobalRaw(orgSeqs.first.data() + rr.first, rr.second, orgSeqs.second.data() + qq.first, qq.second, arena.data() + arenaOff[e], param, sc);
        }
    }
    // output offsets: the path elements before an event's step, then its segment
    std::vector<size_t> off(ev.size() + 1);
    size_t total = n;
    for (size_t e = 0; e < ev.size(); ++e) { off[e] = ev[e].a + (total - n); total += segLen[e]; }
    off[ev.size()] = total;
    const size_t base = after.size();
    after.resize(base + total);
    int8_t *out = after.data() + base;
    for (size_t e = 0; e <= ev.size(); ++e) {      // (a few hundred KB of block copies at most: one thread)
        // the stretch of the path between the previous event's step and this one's (or the end)
        const size_t a0 = e ? ev[e - 1].a : 0, a1 = (e < ev.size()) ? ev[e].a : n;
        const size_t dst = e ? off[e - 1] + segLen[e - 1] : 0;
        if (a1 > a0) std::memcpy(out + dst, before.data() + a0, a1 - a0);
        if (e < ev.size()) {
            int8_t *seg = out + off[e];
            if (ev[e].r >= 0 && ev[e].q >= 0) { if (segLen[e]) std::memcpy(seg, arena.data() + arenaOff[e], segLen[e]); }
            else std::memset(seg, ev[e].r >= 0 ? 2 : 1, segLen[e]);
        }
    }
}

// alignment-helper.cpp:377-503: rewrite every member row with the new gaps; keepCode = the path code that also keeps a letter
static void applyPathToSide(Node *node, SequenceDB *db, const alnPath &aln, int8_t keepCode)
{
    const int total = (int)aln.size();
    const int nSeq = (int)node->seqsIncluded.size();
#pragma omp parallel for schedule(dynamic, 1) if (nSeq > 1)
    for (int k = 0; k < nSeq; ++k) {
        const int sIdx = node->seqsIncluded[k];
        if (db->currentTask != 2 && sIdx >= 0) {
            auto *s = db->sequences[sIdx];
            s->memCheck(total);
            const char *from = s->alnStorage[s->storage];
            char *to = s->alnStorage[1 - s->storage];
            int org = 0;
            for (int c = 0; c < total; ++c) {
                if (aln[c] == 0 || aln[c] == keepCode) to[c] = from[org++];
                else to[c] = '-';
            }
            s->len = total;
            s->changeStorage();
        } else {                                     // a compressed group: compose its stored path instead
            alnPath old;
            { std::lock_guard<std::mutex> lk(g_mapMutex); old = db->subtreeAln[sIdx]; }
            alnPath upd(total);
            int org = 0;
            for (int c = 0; c < total; ++c) {
                if (aln[c] == 0 || aln[c] == keepCode) upd[c] = old[org++];
                else upd[c] = 1;
            }
            { std::lock_guard<std::mutex> lk(g_mapMutex); db->subtreeAln[sIdx] = upd; }
        }
    }
}

void updateAlignment(NodePair &nodes, SequenceDB *database, Option *, alnPath &aln)
{
    const int totalLen = (int)aln.size();
    applyPathToSide(nodes.first, database, aln, 2);
    applyPathToSide(nodes.second, database, aln, 1);
    nodes.first->alnNum += nodes.second->alnNum;
    nodes.first->alnLen = totalLen;
    nodes.first->alnWeight += nodes.second->alnWeight;
    for (int idx : nodes.second->seqsIncluded) nodes.first->seqsIncluded.push_back(idx);
    nodes.second->seqsIncluded.clear();
    if (nodes.first->seqsIncluded.size() > (size_t)database->updateSeqTh && !nodes.first->msaFreq.empty() && database->currentTask != 2) {   // :479-500
        int seqCount = 0, firstSeqID = 0;
        for (int idx : nodes.first->seqsIncluded)
            if (idx > 1) { if (firstSeqID == 0) firstSeqID = -idx; seqCount++; }
        if (seqCount >= database->updateSeqTh) {
            { std::lock_guard<std::mutex> lk(g_mapMutex); database->subtreeAln[firstSeqID] = alnPath(totalLen, 0); }
            std::vector<int> kept{firstSeqID};
            for (int idx : nodes.first->seqsIncluded) {
                if (idx >= 0) database->sequences[idx]->subtreeIdx = firstSeqID;
                else kept.push_back(idx);
            }
            nodes.first->seqsIncluded = kept;
        }
    }
}

// alignment-helper.cpp:506-539
void updateFrequency(NodePair &nodes, SequenceDB *, alnPath &aln, FloatPair weights)
{
    if (nodes.first->msaFreq.empty() || nodes.second->msaFreq.empty()) return;
    const int P = (int)nodes.first->msaFreq[0].size();
    const float refWeight = weights.first, qryWeight = weights.second;
    Profile merged(aln.size(), std::vector<float>(P, 0.0f));
    int r = 0, q = 0;
    for (size_t j = 0; j < aln.size(); ++j) {
        if (aln[j] == 0) {
            for (int k = 0; k < P; ++k) merged[j][k] = nodes.first->msaFreq[r][k] + nodes.second->msaFreq[q][k];
            ++r; ++q;
        } else if (aln[j] == 1) {
            for (int k = 0; k < P - 1; ++k) merged[j][k] = nodes.second->msaFreq[q][k];
            merged[j][P - 1] = nodes.second->msaFreq[q][P - 1] + 1.0 * refWeight;
            ++q;
        } else if (aln[j] == 2) {
            for (int k = 0; k < P - 1; ++k) merged[j][k] = nodes.first->msaFreq[r][k];
            merged[j][P - 1] = nodes.first->msaFreq[r][P - 1] + 1.0 * qryWeight;
            ++r;
        }
    }
    nodes.second->msaFreq.clear();
    nodes.first->msaFreq = std::move(merged);
    nodes.first->alnLen = (int)nodes.first->msaFreq.size();
}

// alignment-helper.cpp:541-591: the smaller / low-quality side of a failed pair is deferred; the kept side ends up in `first`
void fallback2cpu(std::vector<int> &fallbackPairs, NodePairVec &nodes, SequenceDB *database, Option *option)
{
    int totalSeqs = 0;
    const bool filtering = !option->noFilter;
    std::sort(fallbackPairs.begin(), fallbackPairs.end());
    for (int nIdx : fallbackPairs) {
        Node *a = nodes[nIdx].first, *b = nodes[nIdx].second;
        const int32_t refNum = a->alnNum, qryNum = b->alnNum;
        const bool lowQ_r = (refNum > 1) ? false : database->sequences[a->seqsIncluded[0]]->lowQuality;
        const bool lowQ_q = (qryNum > 1) ? false : database->sequences[b->seqsIncluded[0]]->lowQuality;
        if (refNum < qryNum || lowQ_r) {
            if (!filtering || !lowQ_r) {
                database->fallback_nodes.push_back(b);
                if (lowQ_r) database->sequences[a->seqsIncluded[0]]->lowQuality = false;
            }
            std::swap(a->alnLen, b->alnLen);
            std::swap(a->alnNum, b->alnNum);
            std::swap(a->alnWeight, b->alnWeight);
            std::swap(a->seqsIncluded, b->seqsIncluded);
            std::swap(a->msaFreq, b->msaFreq);
            std::swap(a->cacheId, b->cacheId);
            totalSeqs += refNum;
        } else {
            if (!filtering || !lowQ_q) {
                database->fallback_nodes.push_back(b);
                if (lowQ_q) database->sequences[b->seqsIncluded[0]]->lowQuality = false;
            }
            totalSeqs += qryNum;
        }
    }
    if (option->printDetail) printf("Deferring/excluding %lu pair (%d sequences).\n", fallbackPairs.size(), totalSeqs);
}

}  // namespace alignment_helper

namespace progressive {

// alignment-cpu.cpp:50-66,72-93: everything a pair needs before the DP
void preparePair(NodePair &nodes, SequenceDB *database, Option *option, Params &param, PairInputs &in, float *freqSlot, float *gapOpSlot,
                 float *gapExSlot, int stride)
{
    const int P = param.matrixSize + 1;
    in.refLen = nodes.first->getAlnLen(database->currentTask);
    in.qryLen = nodes.second->getAlnLen(database->currentTask);
    in.refNum = nodes.first->getAlnNum(database->currentTask);
    in.qryNum = nodes.second->getAlnNum(database->currentTask);
    in.memLen = std::max(in.refLen, in.qryLen);
    if (freqSlot) {
        if (stride < in.memLen) { std::cerr << "ERROR: staging stride " << stride << " < profile length " << in.memLen << '\n'; exit(1); }
        in.memLen = stride;
        in.freq.bind(freqSlot, (size_t)P * 2 * stride);
        in.gapOp.bind(gapOpSlot, (size_t)2 * stride);
        in.gapEx.bind(gapExSlot, (size_t)2 * stride);
    } else {
        in.freq.assign((size_t)P * 2 * in.memLen, 0.0f);
        in.gapOp.assign((size_t)2 * in.memLen, 0.0f);
        in.gapEx.assign((size_t)2 * in.memLen, 0.0f);
    }
    in.gappyColumns = {};
    in.consensus = {"", ""};
    in.lens = {in.refLen, in.qryLen};
    alignment_helper::calculateProfile(in.freq.data(), nodes, database, option, in.memLen);
    alignment_helper::getConsensus(option, in.freq.data(), in.consensus.first, in.refLen);
    alignment_helper::getConsensus(option, in.freq.data() + (size_t)P * in.memLen, in.consensus.second, in.qryLen);
    alignment_helper::removeGappyColumns(in.freq.data(), nodes, option, in.gappyColumns, in.memLen, in.lens, database->currentTask);
    alignment_helper::calculatePSGP(in.freq.data(), in.gapOp.data(), in.gapEx.data(), nodes, database, option, in.memLen, {0, 0}, in.lens, param);
    in.lowQ_r = (option->alnMode == MERGE_MSA) ? false : ((in.refNum > 1) ? false : database->sequences[nodes.first->seqsIncluded[0]]->lowQuality);
    in.lowQ_q = (option->alnMode == MERGE_MSA) ? false : ((in.qryNum > 1) ? false : database->sequences[nodes.second->seqsIncluded[0]]->lowQuality);
}

// alignment-cpu.cpp:136-175: low-quality rule, gappy columns back, write-back.  aln_wo_gc empty = the DP produced no path.
bool finishPair(NodePair &nodes, SequenceDB *database, Option *option, Params &param, PairInputs &in, alnPath &aln_wo_gc)
{
    if (database->currentTask == 0 && (in.refNum == 1 || in.qryNum == 1) && (in.lowQ_r || in.lowQ_q)) aln_wo_gc.clear();
    if (aln_wo_gc.empty()) return false;
    alnPath aln_w_gc;
    int alnRef = 0, alnQry = 0;
    for (auto a : aln_wo_gc) { if (a != 1) ++alnRef; if (a != 2) ++alnQry; }
    alignment_helper::addGappyColumnsBack(aln_wo_gc, aln_w_gc, in.gappyColumns, param, {alnRef, alnQry}, in.consensus);
    alnRef = alnQry = 0;
    for (auto a : aln_w_gc) { if (a != 1) ++alnRef; if (a != 2) ++alnQry; }
    const float refWeight = nodes.first->alnWeight, qryWeight = nodes.second->alnWeight;
    if (alnRef != in.refLen) std::cout << "R: Post " << nodes.first->identifier << "(" << alnRef << "/" << in.refLen << ")\n";
    if (alnQry != in.qryLen) std::cout << "Q: Post " << nodes.second->identifier << "(" << alnQry << "/" << in.qryLen << ")\n";
    alignment_helper::updateFrequency(nodes, database, aln_w_gc, {refWeight, qryWeight});
    alignment_helper::updateAlignment(nodes, database, option, aln_w_gc);
    return true;
}

}  // namespace progressive
}  // namespace msa

// tests/host_level_dump.cpp -- runs the host mirror's preparePair / finishPair (twilight_amd/csrc/host/helpers.cpp) on one pair
// described in a text file and dumps every intermediate as bit patterns, so that tests can hold oracle/level_oracle.py (numpy)
// against the C++ mirror that the end-to-end pins validate.  Test infrastructure.
//
// input:  type n|p / thr <f> / side <0|1> <alnWeight> <nRows> <hasCache> followed by nRows lines "<weight> <row>" and, when hasCache,
//         one line of len*P hex words (the node's msaFreq) / path <codes>  (aln_wo_gc; "-" = none)
#include "../twilight_amd/csrc/host/twl_host.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

using namespace msa;

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float fromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream in(argv[1]);
    Option opt;
    std::string tok;
    in >> tok >> tok;
    opt.type = tok[0];
    in >> tok >> opt.gappyVertical;
    const int P = (opt.type == 'n') ? 6 : 22;
    Params param(opt, opt.type);
    SequenceDB db;
    Node a("a", 0), b("b", 0);
    Node *nd[2] = {&a, &b};
    int nextId = 0;
    for (int sd = 0; sd < 2; ++sd) {
        int which, nRows, hasCache;
        float alnWeight;
        in >> tok >> which >> alnWeight >> nRows >> hasCache;
        Node *n = nd[which];
        n->alnWeight = alnWeight;
        n->alnNum = nRows;
        for (int r = 0; r < nRows; ++r) {
            float w;
            std::string row;
            in >> w >> row;
            db.addSequence(nextId, "s" + std::to_string(nextId), row, 0, w, false);
            n->seqsIncluded.push_back(nextId++);
            n->alnLen = (int)row.size();
        }
        if (hasCache) {
            n->msaFreq.assign(n->alnLen, std::vector<float>(P));
            for (int t = 0; t < n->alnLen; ++t)
                for (int v = 0; v < P; ++v) { uint32_t u; in >> std::hex >> u >> std::dec; n->msaFreq[t][v] = fromBits(u); }
        }
    }
    std::string pathStr;
    in >> tok >> pathStr;

    NodePair np{&a, &b};
    progressive::PairInputs pi;
    progressive::preparePair(np, &db, &opt, param, pi);
    printf("lens %d %d\n", pi.lens.first, pi.lens.second);
    for (int sd = 0; sd < 2; ++sd) {
        const int L = sd ? pi.lens.second : pi.lens.first;
        printf("cols %d", sd);
        for (int t = 0; t < L; ++t) {
            for (int v = 0; v < P; ++v) printf(" %08x", bits(pi.freq.data()[(size_t)sd * P * pi.memLen + (size_t)P * t + v]));
            printf(" %08x %08x", bits(pi.gapOp.data()[(size_t)sd * pi.memLen + t]), bits(pi.gapEx.data()[(size_t)sd * pi.memLen + t]));
        }
        printf("\nruns %d", sd);
        for (auto &r : (sd ? pi.gappyColumns.second : pi.gappyColumns.first)) printf(" %d,%d", r.first, r.second);
        printf("\ncons %d %s\n", sd, (sd ? pi.consensus.second : pi.consensus.first).c_str());
        printf("cache %d", sd);
        for (auto &col : nd[sd]->msaFreq) for (float f : col) printf(" %08x", bits(f));
        printf("\n");
    }
    if (pathStr != "-") {
        alnPath p;
        for (char c : pathStr) p.push_back((int8_t)(c - '0'));
        // finishPair = addGappyColumnsBack + updateFrequency + updateAlignment; print the full path first
        alnPath full;
        auto gappy = pi.gappyColumns;
        alnPath pcopy = p;
        alignment_helper::addGappyColumnsBack(pcopy, full, gappy, param, {0, 0}, pi.consensus);
        printf("full ");
        for (auto c : full) printf("%d", (int)c);
        printf("\n");
        progressive::finishPair(np, &db, &opt, param, pi, p);
        for (auto *s : db.sequences) printf("row %d %.*s\n", s->id, s->len, s->alnStorage[s->storage]);
        printf("merged");
        for (auto &col : a.msaFreq) for (float f : col) printf(" %08x", bits(f));
        printf("\n");
    }
    return 0;
}

// twilight_amd/csrc/host/seqdb_io.cpp -- sequence storage, scoring parameters, FASTA in / MSA out.
// Mirrors /root/reference/src/sequencedb.cpp, scoring-matrix.cpp, io.cpp (readSequences, writeAlignment) for DEFAULT_ALN.
#include "twl_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

// ---- letters (scoring-matrix.cpp:10-79) -------------------------------------------------------------------
char checkOnly(char c)
{
    switch (c) {
    case 'E': case 'F': case 'I': case 'J': case 'L': case 'P': case 'Q': case 'Z': return 'p';
    case 'U': return 'n';
    default: return 'x';
    }
}

int letterIdx(char type, char c)
{
    if (type == 'p') {
        static const char *aa = "ACDEFGHIKLMNPQRSTVWY";
        if (c == '-' || c == '.') return 21;
        const char *p = c ? strchr(aa, c) : nullptr;
        return p ? (int)(p - aa) : 20;
    }
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': case 'U': return 3;
    case '-': case '.': return 5;
    default: return 4;
    }
}

namespace msa {

// BLOSUM62 in ACDEFGHIKLMNPQRSTVWY order (public NCBI table; reference blosum.hpp holds the same values)
static const int8_t kBlosum62[20][20] = {
    /*A*/ {4, 0, -2, -1, -2, 0, -2, -1, -1, -1, -1, -2, -1, -1, -1, 1, 0, 0, -3, -2},
    /*C*/ {0, 9, -3, -4, -2, -3, -3, -1, -3, -1, -1, -3, -3, -3, -3, -1, -1, -1, -2, -2},
    /*D*/ {-2, -3, 6, 2, -3, -1, -1, -3, -1, -4, -3, 1, -1, 0, -2, 0, -1, -3, -4, -3},
    /*E*/ {-1, -4, 2, 5, -3, -2, 0, -3, 1, -3, -2, 0, -1, 2, 0, 0, -1, -2, -3, -2},
    /*F*/ {-2, -2, -3, -3, 6, -3, -1, 0, -3, 0, 0, -3, -4, -3, -3, -2, -2, -1, 1, 3},
    /*G*/ {0, -3, -1, -2, -3, 6, -2, -4, -2, -4, -3, 0, -2, -2, -2, 0, -2, -3, -2, -3},
    /*H*/ {-2, -3, -1, 0, -1, -2, 8, -3, -1, -3, -2, 1, -2, 0, 0, -1, -2, -3, -2, 2},
    /*I*/ {-1, -1, -3, -3, 0, -4, -3, 4, -3, 2, 1, -3, -3, -3, -3, -2, -1, 3, -3, -1},
    /*K*/ {-1, -3, -1, 1, -3, -2, -1, -3, 5, -2, -1, 0, -1, 1, 2, 0, -1, -2, -3, -2},
    /*L*/ {-1, -1, -4, -3, 0, -4, -3, 2, -2, 4, 2, -3, -3, -2, -2, -2, -1, 1, -2, -1},
    /*M*/ {-1, -1, -3, -2, 0, -3, -2, 1, -1, 2, 5, -2, -2, 0, -1, -1, -1, 1, -1, -1},
    /*N*/ {-2, -3, 1, 0, -3, 0, 1, -3, 0, -3, -2, 6, -2, 0, 0, 1, 0, -3, -4, -2},
    /*P*/ {-1, -3, -1, -1, -4, -2, -2, -3, -1, -3, -2, -2, 7, -1, -2, -1, -1, -2, -4, -3},
    /*Q*/ {-1, -3, 0, 2, -3, -2, 0, -3, 1, -2, 0, 0, -1, 5, 1, 0, -1, -2, -2, -1},
    /*R*/ {-1, -3, -2, 0, -3, -2, 0, -3, 2, -2, -1, 0, -2, 1, 5, -1, -1, -3, -3, -2},
    /*S*/ {1, -1, 0, 0, -2, 0, -1, -2, 0, -2, -1, 1, -1, 0, -1, 4, 1, -2, -3, -2},
    /*T*/ {0, -1, -1, -1, -2, -2, -2, -1, -1, -1, -1, 0, -1, -1, -1, 1, 5, 0, -2, -2},
    /*V*/ {0, -1, -3, -2, -1, -3, -3, 3, -2, 1, 1, -3, -2, -2, -3, -2, 0, 4, -3, -1},
    /*W*/ {-3, -2, -4, -3, 1, -2, -2, -3, -3, -2, -1, -4, -4, -2, -3, -3, -2, -3, 11, 2},
    /*Y*/ {-2, -2, -3, -2, 3, -3, 2, -1, -2, -1, -1, -2, -3, -1, -2, -2, -2, -1, 2, 7}};

// scoring-matrix.cpp:81-135 (built-in matrices; BLOSUM62 only for proteins in this build)
Params::Params(const Option &o, char type)
{
    gapOpen = o.gapOpen;
    gapExtend = o.gapExtend;
    gapBoundary = o.hasGapEnds ? o.gapEnds : gapExtend;
    float xd = std::round(o.xdrop);
    if (gapOpen > 0 || gapExtend > 0 || gapBoundary > 0) { std::cerr << "ERROR: Gap penalties must be less than or equal to 0.\n"; exit(1); }
    if (xd <= 0) { std::cerr << "ERROR: XDrop value should be larger than 0.\n"; exit(1); }
    xdrop = (gapExtend == 0) ? xd : -1 * xd * gapExtend;
    matrixSize = (type == 'n') ? 5 : 21;
    scoringMatrix = new float *[matrixSize];
    for (int i = 0; i < matrixSize; ++i) scoringMatrix[i] = new float[matrixSize];
    if (type == 'n') {
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                if (i == 4 || j == 4) scoringMatrix[i][j] = o.wildcard ? o.match : 0.0f;
                else if (i == j) scoringMatrix[i][j] = o.match;
                else if (std::abs(i - j) == 2) scoringMatrix[i][j] = o.transition;
                else scoringMatrix[i][j] = o.mismatch;
            }
    } else {
        if (o.blosum != 62) std::cerr << "WARNING: only BLOSUM62 is built in; using BLOSUM62.\n";
        float Nscore = 0;
        for (int i = 0; i < 20; ++i) Nscore += kBlosum62[i][i];
        Nscore /= 20;
        for (int i = 0; i < 21; ++i) {
            scoringMatrix[i][20] = o.wildcard ? 5 * Nscore : 0.0f;
            scoringMatrix[20][i] = o.wildcard ? 5 * Nscore : 0.0f;
        }
        for (int i = 0; i < 20; ++i)
            for (int j = 0; j < 20; ++j) scoringMatrix[i][j] = 5 * kBlosum62[i][j];
    }
}

Params::~Params()
{
    for (int i = 0; i < matrixSize; ++i) delete[] scoringMatrix[i];
    delete[] scoringMatrix;
}

// ---- SequenceDB (sequencedb.cpp:8-85) ---------------------------------------------------------------------
SequenceDB::SequenceInfo::SequenceInfo(int id_, const std::string &name_, std::string &seq, int subtreeIdx_, float weight_, bool debug)
    : id(id_), name(name_), len((int)seq.length()), subtreeIdx(subtreeIdx_), weight(weight_)
{
    memLen = len * timesBigger;
    // calloc: the spare capacity and the second buffer stay untouched (no page is faulted in) until a level writes there
    alnStorage[0] = static_cast<char *>(calloc(memLen > 0 ? memLen : 1, 1));
    alnStorage[1] = static_cast<char *>(calloc(memLen > 0 ? memLen : 1, 1));
    if (len > 0) memcpy(alnStorage[0], seq.data(), (size_t)len);
    if (debug) unalignedSeq = seq;
}

SequenceDB::SequenceInfo::~SequenceInfo()
{
    if (!borrowed) {
        free(alnStorage[0]);
        free(alnStorage[1]);
    }
}

void SequenceDB::SequenceInfo::memCheck(int need)
{
    if (memLen >= need) return;
    const int grown = need * timesBigger;
    for (int b = 0; b < 2; ++b) {
        char *t = static_cast<char *>(calloc(grown, 1));
        if (memLen > 0) memcpy(t, alnStorage[b], (size_t)memLen);
        if (!borrowed) free(alnStorage[b]);
        alnStorage[b] = t;
    }
    borrowed = false;
    memLen = grown;
}

void SequenceDB::addSequence(int id, const std::string &name, std::string &seq, int subtreeIdx, float weight, bool debug)
{
    SequenceInfo *s = new SequenceInfo(id, name, seq, subtreeIdx, weight, debug);
    sequences.push_back(s);
    name_map[name] = s;
}

SequenceDB::~SequenceDB()
{
    if (gpuCtx && gpuCtxFree) gpuCtxFree(gpuCtx);
    gpuCtx = nullptr;
    for (auto *s : sequences) delete s;
    free(rowArena);
}

// sequencedb.cpp:87-120 (--check): legality of the MSA, not optimality
bool SequenceDB::debug()
{
    bool ok = true, first = true;
    int alnLen = 0, checked = 0;
    for (auto *s : sequences) {
        if (s->lowQuality) continue;
        const char *row = s->alnStorage[s->storage];
        std::string r;
        int off = 0;
        while (off < s->memLen && (isalpha((unsigned char)row[off]) || row[off] == '-' || row[off] == '.')) {
            if (row[off] != '-' && row[off] != '.') r += row[off];
            ++off;
        }
        if (first) { alnLen = off; first = false; }
        else if (alnLen != off) { printf("%s: the sequence length (%d) did not match the MSA length(%d)\n", s->name.c_str(), off, alnLen); ok = false; }
        if (r != s->unalignedSeq) { printf("%s: after removing the gaps, the alignment did not match the original sequence.\n", s->name.c_str()); ok = false; }
        ++checked;
    }
    std::cerr << "Completed checking " << checked << " sequences.\n";
    return ok;
}

namespace io {

// FASTA(.gz) records the way kseq.h delivers them: name = header up to the first blank, sequence = all lines joined.
// The file is pulled through zlib in 4 MB blocks (gzread handles plain files transparently) and split with memchr.
struct ChunkReader {
    gzFile f;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    explicit ChunkReader(gzFile f_) : f(f_), buf(1 << 22) {}
    bool fill()
    {
        if (eof) return false;
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        pos = 0;
        end = n > 0 ? (size_t)n : 0;
        if (n <= 0) eof = true;
        return n > 0;
    }
    // Appends the next physical line (without the newline) to `line`; false at end of file with nothing read.
    bool getline(std::string &line)
    {
        bool any = false;
        for (;;) {
            if (pos == end && !fill()) return any;
            const char *s = buf.data() + pos;
            const char *nl = static_cast<const char *>(memchr(s, '\n', end - pos));
            if (nl) { line.append(s, nl - s); pos += (size_t)(nl - s) + 1; return true; }
            line.append(s, end - pos);
            pos = end;
            any = true;
        }
    }
};

static bool nextRecord(ChunkReader &in, std::string &carry, std::string &name, std::string &seq)
{
    name.clear();
    seq.clear();
    bool have = false;
    auto header = [&](const std::string &line) {
        size_t e = line.find_first_of(" \t", 1);
        name = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
        have = true;
    };
    if (!carry.empty()) { header(carry); carry.clear(); }
    std::string line;
    while (true) {
        line.clear();
        if (!in.getline(line)) return have;
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (have) { carry = line; return true; }
            header(line);
        } else if (have) {
            const size_t at = seq.size();
            seq += line;                               // whole line at once; blanks inside a sequence line are rare
            if (line.find_first_of(" \t\v\f\r\n") != std::string::npos)
                seq.erase(std::remove_if(seq.begin() + at, seq.end(), [](char c) { return isspace((unsigned char)c) != 0; }), seq.end());
        }
    }
}

char detectType(const std::string &seqFile)         // option.cpp:115-171: first decisive letter in the first 100 sequence lines
{
    gzFile f = gzopen(seqFile.c_str(), "r");
    if (!f) { std::cerr << "ERROR: Failed to open file " << seqFile << ".\n"; exit(1); }
    char buf[4096];
    char type = 'n';
    int lines = 0;
    while (gzgets(f, buf, sizeof buf)) {
        if (buf[0] == '\0' || buf[0] == '\n' || buf[0] == '>') continue;
        bool fin = false;
        for (char *p = buf; *p; ++p) {
            char t = checkOnly((char)toupper((unsigned char)*p));
            if (t != 'x') { type = t; fin = true; break; }
        }
        if (fin || ++lines == 100) break;
    }
    gzclose(f);
    return type;
}

// io.cpp:55-198
void readSequences(const std::string &fileName, SequenceDB *database, Option *option, Tree *&tree)
{
    gzFile f = gzopen(fileName.c_str(), "r");
    if (!f) { fprintf(stderr, "ERROR: cant open file: %s\n", fileName.c_str()); exit(1); }
    gzbuffer(f, 1 << 22);
    ChunkReader in(f);
    const int seqNum_init = (int)database->sequences.size();
    int seqNum = seqNum_init, maxLen = 0, minLen = INT_MAX;
    uint64_t totalLen = 0;
    std::vector<int> lens;
    std::string carry, name, seq;
    while (nextRecord(in, carry, name, seq)) {
        if (tree->allNodes.find(name) == tree->allNodes.end()) continue;
        if (database->name_map.count(name)) {
            printf("WARNING: duplicate leaf names found in the sequence file! Leaf name: %s. Only the first occurrence will be kept.\n", name.c_str());
            continue;
        }
        const int L = (int)seq.size();
        maxLen = std::max(maxLen, L);
        minLen = std::min(minLen, L);
        if (L == 0) std::cerr << "Null sequences, " << name << '\n';
        totalLen += L;
        Node *leaf = tree->allNodes[name];
        database->addSequence(seqNum, name, seq, leaf->grpID, leaf->weight, option->debug);
        leaf->placed = false;
        ++seqNum;
        lens.push_back(L);
    }
    gzclose(f);

    if (tree->m_numLeaves != (size_t)seqNum && option->alnMode == DEFAULT_ALN) {      // io.cpp:102-118
        printf("Warning: Mismatch between the number of leaves and the number of sequences, (%lu != %d)\n", tree->m_numLeaves, seqNum);
        for (auto &kv : tree->allNodes)
            if (kv.second->is_leaf() && !database->name_map.count(kv.second->identifier)) std::cerr << "Missing " << kv.second->identifier << '\n';
        std::cerr << "Prune the tree according to the existing sequences.\n";
        std::unordered_set<std::string> names;
        for (auto &kv : database->name_map) names.insert(kv.first);
        phylogeny::pruneTree(tree, names);
    }
    if (seqNum == seqNum_init) { std::cerr << "Error: no sequences were read from the input.\n"; exit(1); }

    std::sort(lens.begin(), lens.end());
    const uint32_t avgLen = (uint32_t)(totalLen / (seqNum - seqNum_init));
    const uint32_t medLen = lens[(seqNum - seqNum_init) / 2];
    const int minTh = (option->lenDev > 0) ? (int)(medLen * (1 - option->lenDev)) : option->minLen;
    const int maxTh = (option->lenDev > 0) ? (int)(medLen * (1 + option->lenDev)) : option->maxLen;
    std::atomic<int> numLowQ{0};
    uint8_t isAmbig[256];                     // letterIdx(type, toupper(c)) == N / X, tabulated once
    for (int c = 0; c < 256; ++c) isAmbig[c] = (letterIdx(option->type, (char)toupper(c)) == ((option->type == 'n') ? 4 : 20)) ? 1 : 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < seqNum; ++i) {                                                  // io.cpp:134-162
        auto *s = database->sequences[i];
        s->lowQuality = (s->len > maxTh || s->len < minTh);
        if (!s->lowQuality) {
            int cnt = 0;
            const unsigned char *row = reinterpret_cast<const unsigned char *>(s->alnStorage[0]);
            for (int j = 0; j < s->len; ++j) cnt += isAmbig[row[j]];
            s->lowQuality = (cnt > (s->len * option->maxAmbig));
        }
        if (s->lowQuality) {
            numLowQ.fetch_add(1);
            if (!option->noFilter) s->len = 0;
        }
    }
    std::cerr << "===== Sequence Summary =====\nNumber : " << (seqNum - seqNum_init) << "\nMax. Length: " << maxLen << "\nMin. Length: " << minLen
              << "\nAvg. Length: " << avgLen << "\nMed. Length: " << medLen << '\n'
              << (option->noFilter ? "Deferred sequences: " : "Excluded sequences: ") << numLowQ << '\n';
}

void writeAlignment(const std::string &fileName, SequenceDB *database, int alnLen)     // io.cpp:512-525 (plain FASTA, input order)
{
    std::ofstream out(fileName, std::ios::binary);
    if (!out) { fprintf(stderr, "ERROR: Failed to open file: %s\n", fileName.c_str()); exit(1); }
    for (auto *s : database->sequences) {
        if (s->lowQuality) continue;
        out << '>' << s->name << "\n";
        out.write(&s->alnStorage[s->storage][0], alnLen);
        out << '\n';
    }
}

void writeFinalMSA(SequenceDB *database, Option *option, int alnLen)                  // io.cpp:465-488, DEFAULT_ALN branch
{
    std::cerr << "Final Alignment Length: " << alnLen << '\n';
    writeAlignment(option->outFile, database, alnLen);
}

}  // namespace io
}  // namespace msa

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

// The built-in substitution tables, ACDEFGHIKLMNPQRSTVWY order, as the reference ships them (blosum.hpp:9-79; data, not code).
// BLOSUM80 is NOT symmetric there: [V][I] = 1 but [I][V] = 3 (blosum.hpp:65,75), which makes the DP's matrix indexing
// scoreMatrix[l][m] with l = reference letter, m = query letter (TALCO-XDrop.cpp:382) observable.
static const int8_t kBlosum45[20][20] = {
    /*A*/ {5, -1, -2, -1, -2, 0, -2, -1, -1, -1, -1, -1, -1, -1, -2, 1, 0, 0, -2, -2},
    /*C*/ {-1, 12, -3, -3, -3, -3, -3, -3, -2, -2, -2, -2, -4, -2, -3, -3, -1, -1, -5, -3},
    /*D*/ {-2, -3, 7, 2, -4, -1, 0, -4, 0, -3, -3, 2, -1, 0, -1, -1, -1, -3, -4, -3},
    /*E*/ {-1, -3, 2, 6, -3, -2, 0, -3, 1, -2, -2, 0, -1, 2, 0, 0, 0, -2, -3, -3},
    /*F*/ {-2, -3, -4, -3, 8, -3, 0, 0, -3, 1, 1, -2, -4, -2, -3, -2, -2, -1, 0, 1},
    /*G*/ {0, -3, -1, -2, -3, 7, -2, -2, -2, -3, -2, -1, -2, -2, -2, 0, -1, -2, -2, -2},
    /*H*/ {-2, -3, 0, 0, 0, -2, 10, -3, -1, -2, -2, 1, 1, -3, 1, 0, -1, -2, -3, -2},
    /*I*/ {-1, -3, -4, -3, 0, -2, -3, 5, -3, 5, 2, -2, -3, -2, -3, -3, -2, 3, -2, -2},
    /*K*/ {-1, -2, 0, 1, -3, -2, -1, -3, 5, -3, -1, 0, -1, 0, 1, 3, -1, -2, -2, -2},
    /*L*/ {-1, -2, -3, -2, 1, -3, -2, 5, -3, 5, 3, -3, -2, -3, -2, -2, -2, 1, -2, -1},
    /*M*/ {-1, -2, -3, -2, 1, -2, -2, 2, -1, 3, 6, -3, -2, -2, -1, -2, -2, 1, -1, -1},
    /*N*/ {-1, -2, 2, 0, -2, -1, 1, -2, 0, -3, -3, 6, -2, -2, -2, 1, 0, -3, -4, -2},
    /*P*/ {-1, -4, -1, -1, -4, -2, 1, -3, -1, -2, -2, -2, 9, -2, -2, -1, -1, -2, -3, -3},
    /*Q*/ {-1, -2, 0, 2, -2, -2, -3, -2, 0, -3, -2, -2, -2, 6, 2, 0, -1, -2, -2, -2},
    /*R*/ {-2, -3, -1, 0, -3, -2, 1, -3, 1, -2, -1, -2, -2, 2, 7, -1, -1, -3, -2, -2},
    /*S*/ {1, -3, -1, 0, -2, 0, 0, -3, 3, -2, -2, 1, -1, 0, -1, 4, 2, -2, -4, -2},
    /*T*/ {0, -1, -1, 0, -2, -1, -1, -2, -1, -2, -2, 0, -1, -1, -1, 2, 5, 0, -3, -1},
    /*V*/ {0, -1, -3, -2, -1, -2, -2, 3, -2, 1, 1, -3, -2, -2, -3, -2, 0, 5, -3, -1},
    /*W*/ {-2, -5, -4, -3, 0, -2, -3, -2, -2, -2, -1, -4, -3, -2, -2, -4, -3, -3, 15, 3},
    /*Y*/ {-2, -3, -3, -3, 1, -2, -2, -2, -2, -1, -1, -2, -3, -2, -2, -2, -1, -1, 3, 8}};
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
static const int8_t kBlosum80[20][20] = {
    /*A*/ {7, -1, -3, -3, -4, -1, -4, -2, -1, -2, -2, -3, -3, -2, -3, 1, 0, -1, -6, -4},
    /*C*/ {-1, 13, -6, -7, -3, -4, -5, -3, -5, -3, -3, -5, -4, -5, -5, -1, -1, -3, -5, -4},
    /*D*/ {-3, -6, 10, 1, -7, -3, -1, -7, -1, -7, -6, 2, -5, 0, -4, -1, -2, -6, -8, -7},
    /*E*/ {-3, -7, 1, 7, -6, -4, 0, -5, 1, -5, -4, -1, -3, 2, -1, -2, -3, -4, -8, -6},
    /*F*/ {-4, -3, -7, -6, 8, -5, -4, 0, -6, 1, 0, -6, -6, -5, -5, -4, -3, -1, 0, 4},
    /*G*/ {-1, -4, -3, -4, -5, 8, -4, -6, -3, -6, -5, -2, -5, -4, -5, -2, -3, -5, -7, -6},
    /*H*/ {-4, -5, -1, 0, -4, -4, 12, -6, -1, -5, -3, 1, -3, 1, 0, -2, -3, -5, -4, 2},
    /*I*/ {-2, -3, -7, -5, 0, -6, -6, 5, -5, 2, 2, -6, -5, -5, -5, -4, -2, 3, -5, -4},
    /*K*/ {-1, -5, -1, 1, -6, -3, -1, -5, 8, -4, -3, 0, -2, 2, 1, -1, -1, -4, -6, -4},
    /*L*/ {-2, -3, -7, -5, 1, -6, -5, 2, -4, 5, 3, -6, -4, -4, -4, -3, -2, 1, -4, -3},
    /*M*/ {-2, -3, -6, -4, 0, -5, -3, 2, -3, 3, 7, -4, -4, -2, -3, -3, -1, 1, -4, -3},
    /*N*/ {-3, -5, 2, -1, -6, -2, 1, -6, 0, -6, -4, 9, -4, 0, -1, 0, -1, -5, -7, -6},
    /*P*/ {-3, -4, -5, -3, -6, -5, -3, -5, -2, -4, -4, -4, 10, -3, -3, -2, -3, -4, -7, -6},
    /*Q*/ {-2, -5, 0, 2, -5, -4, 1, -5, 2, -4, -2, 0, -3, 8, 1, -1, -1, -4, -7, -4},
    /*R*/ {-3, -5, -4, -1, -5, -5, 0, -5, 1, -4, -3, -1, -3, 1, 8, -1, -1, -4, -7, -6},
    /*S*/ {1, -1, -1, -2, -4, -2, -2, -4, -1, -3, -3, 0, -2, -1, -1, 6, 2, -1, -6, -4},
    /*T*/ {0, -1, -2, -3, -3, -3, -3, -2, -1, -2, -1, -1, -3, -1, -1, 2, 7, 0, -6, -3},
    /*V*/ {-1, -3, -6, -4, -1, -5, -5, 1, -4, 1, 1, -5, -4, -4, -4, -1, 0, 6, -5, -4},
    /*W*/ {-6, -5, -8, -8, 0, -7, -4, -5, -6, -4, -4, -7, -7, -7, -7, -6, -6, -5, 15, 3},
    /*Y*/ {-4, -4, -7, -6, 4, -6, 2, -4, -4, -3, -3, -6, -6, -4, -6, -4, -3, -4, 3, 9}};

// One letter of a user matrix file -> matrix index (scoring-matrix.cpp:26-79)
static int matrixLetter(char type, const std::string &word) { return letterIdx(type, (char)toupper((unsigned char)word[0])); }

static bool isNumberToken(const std::string &w)
{
    try {
        size_t pos = 0;
        (void)std::stod(w, &pos);
        return pos == w.size();
    } catch (...) {
        return false;
    }
}

// scoring-matrix.cpp:81-199.  Built-in: nucleotide match / transition / mismatch with a zero (or, with -w, match) N row and column;
// 5 x BLOSUM45/62/80 with a zero (or, with -w, 5 x the mean BLOSUM62 diagonal) X row and column.  User matrix (-x file): the
// letters first -- matrixSize-1 of them, or matrixSize when the token after them is not a number (then the last letter is the
// ambiguity letter) -- then the scores row by row in the order of those letters; without an ambiguity letter its row and column
// are 0 or, with -w, the mean of the diagonal.
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
    for (int i = 0; i < matrixSize; ++i) { scoringMatrix[i] = new float[matrixSize]; std::fill(scoringMatrix[i], scoringMatrix[i] + matrixSize, 0.0f); }
    if (!o.matrixFile.empty()) {
        std::ifstream in(o.matrixFile);
        if (!in) { fprintf(stderr, "ERROR: can't open %s\n", o.matrixFile.c_str()); exit(1); }
        std::vector<std::string> tok;
        for (std::string w; in >> w;) tok.push_back(w);
        int nLetters = matrixSize - 1;
        if ((int)tok.size() > nLetters && !isNumberToken(tok[nLetters])) nLetters = matrixSize;      // the ambiguity letter is listed too
        const int ambig = matrixSize - 1;
        std::vector<int> idx;
        for (int t = 0; t < nLetters && t < (int)tok.size(); ++t) {
            const int li = matrixLetter(type, tok[t]);
            if (li == ambig && t < matrixSize - 1 && nLetters == matrixSize - 1) {
                std::cerr << "Unrecognized letter \"" << (char)toupper((unsigned char)tok[t][0]) << "\"" << (type == 'n' ? " for nucleotide sequences.\n" : " for protein sequences.\n");
                exit(1);
            }
            idx.push_back(li);
        }
        for (size_t t = (size_t)nLetters; t < tok.size(); ++t) {
            const size_t k = t - (size_t)nLetters;
            const size_t x = k / (size_t)nLetters, y = k % (size_t)nLetters;
            if (x >= idx.size()) break;
            scoringMatrix[idx[x]][idx[y]] = std::stof(tok[t]);
        }
        if (nLetters == matrixSize - 1) {
            float Nscore = 0;
            for (int i = 0; i < nLetters; ++i) Nscore += scoringMatrix[i][i];
            Nscore = o.wildcard ? (Nscore / nLetters) : 0.0f;
            for (int i = 0; i < matrixSize; ++i) { scoringMatrix[i][matrixSize - 1] = Nscore; scoringMatrix[matrixSize - 1][i] = Nscore; }
        }
    } else if (type == 'n') {
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                if (i == 4 || j == 4) scoringMatrix[i][j] = o.wildcard ? o.match : 0.0f;
                else if (i == j) scoringMatrix[i][j] = o.match;
                else if (std::abs(i - j) == 2) scoringMatrix[i][j] = o.transition;
                else scoringMatrix[i][j] = o.mismatch;
            }
    } else {
        int blosumType = o.blosum;
        if (blosumType != 45 && blosumType != 62 && blosumType != 80) {
            std::cerr << "WARNING: Invalid BLOSUM matrix \"" << blosumType << "\". Please choose from 45, 62, or 80.\nUsing default: BLOSUM62.\n";
            blosumType = 62;
        }
        float Nscore = 0;
        for (int i = 0; i < 20; ++i) Nscore += kBlosum62[i][i];      // (the reference takes the BLOSUM62 diagonal whichever table is selected, :120-122)
        Nscore /= 20;
        for (int i = 0; i < 21; ++i) {
            scoringMatrix[i][20] = o.wildcard ? 5 * Nscore : 0.0f;
            scoringMatrix[20][i] = o.wildcard ? 5 * Nscore : 0.0f;
        }
        const int8_t(*tab)[20] = (blosumType == 45) ? kBlosum45 : (blosumType == 80) ? kBlosum80 : kBlosum62;
        for (int i = 0; i < 20; ++i)
            for (int j = 0; j < 20; ++j) scoringMatrix[i][j] = 5.0f * tab[i][j];
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

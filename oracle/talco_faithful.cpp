// oracle/talco_faithful.cpp -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The same algorithm as oracle/talco_oracle.c (Talco_xdrop::Align_freq / Tile / Traceback, /root/reference/src/TALCO-XDrop.cpp:62-108,
// :233-689, :134-231) restated once more with the REFERENCE'S DATA LAYOUT AND ALLOCATION PATTERN, so that bench.py's CPU leg can
// time "what the reference's own code costs per band cell" on the box it runs on instead of quoting a figure from another machine
// (the "faithful" mode):
//   * profiles as std::vector<std::vector<float>> built per pair from the flat arrays (alignment-cpu.cpp:70-86 builds them the same way),
//     gap penalties as std::vector<std::vector<float>>;
//   * 14 new[] / delete[] per tile for the rotating rows (S, CS x3; I, D, CI, CD x2), each initialised element by element (:287-311);
//   * traceback pointers and the per-diagonal bookkeeping in std::vector with push_back (:280-283, :340-344, :556);
//   * the column score with AVX2 masked loads, one 8-lane round per matrix row, stored to a stack array and summed (:378-393; protein
//     :409-430), i.e. the x86 TALCO_SIMD build (CMakeLists.txt:24-27).
// It is NOT the checker (that is talco_oracle.c): tests/test_oracle_cpu.py holds it to the checker bit for bit, bench.py times it.
#include "talco_oracle.h"

#include <immintrin.h>
#include <omp.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace {

constexpr int kIB = -2, kDB = -3;      // TALCO-XDrop.cpp:33-34

struct Faithful {
    const twlo_params *p;
    const std::vector<std::vector<float>> &reference, &query, &gapOp, &gapEx;
    float refNum, qryNum;
    uint64_t cells = 0;
};

inline int32_t uniformOrMinus1(const int32_t *C, int32_t start, int32_t length)      // Reduction_tree, :110-119
{
    int32_t conv = C[start];
    for (int32_t i = start + 1; i <= start + length; ++i)
        if (C[i] != conv) return -1;
    return conv;
}

void tracebackF(const std::vector<int32_t> &ftrLen, const std::vector<int32_t> &ftrLow, int32_t startAddr, int32_t startFtr, int8_t startState,
                int32_t startQ, int32_t startR, const std::vector<int8_t> &tb, std::vector<int8_t> &out, bool firstTile)      // :134-231
{
    int32_t addr = startAddr;
    int32_t ftr = (int16_t)startFtr, idx = (int16_t)startQ, qidx = (int16_t)startQ, ridx = (int16_t)startR;
    int8_t state = startState;
    while (ftr >= 0) {
        const int8_t v = (addr >= 0 && (size_t)addr < tb.size()) ? tb[(size_t)addr] : 0;
        int8_t dir;
        if (state == 0) {
            state = v & 0x03;
            if (state == 0) dir = 0;
            else if (state == 1) { dir = 1; state = (v & 0x04) ? 1 : 0; }
            else { dir = 2; state = (v & 0x08) ? 2 : 0; }
        } else if (state == 1) { dir = 1; state = (v & 0x04) ? 1 : 0; }
        else { dir = 2; state = (v & 0x08) ? 2 : 0; }
        if (ftr > 0) addr = addr - (idx - ftrLow[ftr] + 1) - ftrLen[ftr - 1];
        if (dir == 0) { if (ftr > 1) addr = addr - ftrLen[ftr - 2] + (idx - ftrLow[ftr - 2]); ftr -= 2; idx -= 1; qidx--; ridx--; }
        else if (dir == 1) { if (ftr > 0) addr = addr + (idx - ftrLow[ftr - 1]); ftr -= 1; idx -= 1; qidx--; }
        else { if (ftr > 0) addr = addr + (idx - ftrLow[ftr - 1] + 1); ftr -= 1; ridx--; }
        out.push_back(dir);
        if (firstTile && (ridx < 0 || qidx < 0)) break;
    }
    if (firstTile) {
        while (ridx > -1) { out.push_back(2); ridx--; }
        while (qidx > -1) { out.push_back(1); qidx--; }
    }
}

// :373-444, the TALCO_SIMD branches
inline float columnScore(const twlo_params *p, const float *refColumns, const float *qryColumns, float denominator)
{
    float numerator = 0.0f;
    const float gc = p->gap_char;
    if (p->P == 6) {
        const __m256i mask = _mm256_setr_epi32(-1, -1, -1, -1, -1, 0, 0, 0);
        for (int l = 0; l < 5; ++l) {
            __m256 sumvec = _mm256_setzero_ps();
            const __m256 refv = _mm256_set1_ps(refColumns[l]);
            const __m256 q = _mm256_maskload_ps(qryColumns, mask);
            const __m256 mat = _mm256_maskload_ps(p->matrix + 5 * l, mask);
            __m256 prod = _mm256_mul_ps(q, mat);
            prod = _mm256_mul_ps(prod, refv);
            sumvec = _mm256_add_ps(sumvec, prod);
            alignas(32) float tmp[8];
            _mm256_store_ps(tmp, sumvec);
            numerator += (tmp[0] + tmp[1] + tmp[2] + tmp[3] + tmp[4]);
        }
        for (int l = 0; l < 5; ++l) numerator += refColumns[l] * qryColumns[5] * gc;
        for (int m = 0; m < 5; ++m) numerator += refColumns[5] * qryColumns[m] * gc;
    } else {
        for (int l = 0; l < 21; ++l) {
            __m256 sumvec = _mm256_setzero_ps();
            const float ref_l = refColumns[l];
            const __m256 refv = _mm256_set1_ps(ref_l);
            const float *smat = p->matrix + 21 * l;
            for (int m = 0; m < 16; m += 8) {
                const __m256 q = _mm256_loadu_ps(qryColumns + m);
                const __m256 mat = _mm256_loadu_ps(smat + m);
                __m256 prod = _mm256_mul_ps(q, mat);
                prod = _mm256_mul_ps(prod, refv);
                sumvec = _mm256_add_ps(sumvec, prod);
            }
            for (int m = 16; m < 21; ++m) numerator += ref_l * qryColumns[m] * smat[m];
            alignas(32) float tmp[8];
            _mm256_store_ps(tmp, sumvec);
            numerator += (tmp[0] + tmp[1] + tmp[2] + tmp[3] + tmp[4] + tmp[5] + tmp[6] + tmp[7]);
        }
        for (int l = 0; l < 21; ++l) numerator += refColumns[l] * qryColumns[21] * gc;
        for (int m = 0; m < 21; ++m) numerator += refColumns[21] * qryColumns[m] * gc;
    }
    return numerator / denominator;
}

// :233-689
void tileF(Faithful &c, int32_t &reference_idx, int32_t &query_idx, std::vector<int8_t> &aln, bool &last_tile, int tile, int16_t &errorType)
{
    const twlo_params *p = c.p;
    const float inf = 2.0 * p->xdrop + 1.0;
    const int32_t marker = p->marker;
    bool converged = false, conv_logic = false;
    int32_t reference_length = (int32_t)c.reference.size() - reference_idx, query_length = (int32_t)c.query.size() - query_idx;
    const int32_t fLen = std::min(p->flen, std::min(reference_length, query_length));
    float max_score = 0, max_score_prime = -inf, conv_score = 0;
    int32_t conv_value = 0, conv_ref_idx = 0, conv_query_idx = 0, tb_start_addr = 0, tb_start_ftr = 0, tb_state = 0;
    const float denominator = c.refNum * c.qryNum;
    if (reference_length < 0 || query_length < 0) { errorType = 3; aln.clear(); return; }

    int32_t L[3], U[3];
    float *S[3], *I[2], *D[2];
    int32_t *CS[3], *CI[2], *CD[2];
    const size_t rowlen = (size_t)std::max(fLen, 1) + 1;      // (+1: the guard element of the checker, so that the same reads stay in bounds)
    for (int s = 0; s < 3; ++s) {                               // :287-298: 14 allocations
        S[s] = new float[rowlen]; CS[s] = new int32_t[rowlen];
        if (s < 2) { I[s] = new float[rowlen]; D[s] = new float[rowlen]; CI[s] = new int32_t[rowlen]; CD[s] = new int32_t[rowlen]; }
        L[s] = s; U[s] = -s;
    }
    for (int s = 0; s < 3; ++s)                                 // :300-311
        for (size_t t = 0; t < rowlen; ++t) {
            S[s][t] = -1; CS[s][t] = -1;
            if (s < 2) { I[s][t] = -1; D[s][t] = -1; CI[s][t] = kIB; CD[s][t] = kDB; }
        }
    auto freeMem = [&] {
        for (int s = 0; s < 3; ++s) { delete[] S[s]; delete[] CS[s]; if (s < 2) { delete[] I[s]; delete[] D[s]; delete[] CI[s]; delete[] CD[s]; } }
    };
    std::vector<int8_t> tb;
    std::vector<int32_t> ftr_length, ftr_lower_limit;
    int32_t ftr_addr = 0, last_k = 0, prev_conv_s = -1;

    for (int32_t k = 0; k < reference_length + query_length - 1; ++k) {
        const int c0 = k % 3, c1 = (k + 2) % 3, c2 = (k + 1) % 3, b0 = k % 2, b1 = (k + 1) % 2;
        const int32_t Lk = L[c0], Uk = U[c0];
        if (Lk >= Uk + 1) { last_tile = true; errorType = 1; aln.clear(); freeMem(); return; }
        if (Uk - Lk + 1 > fLen) { last_tile = true; errorType = 2; aln.clear(); freeMem(); return; }
        if (k <= marker) { ftr_length.push_back(Uk - Lk + 1); ftr_lower_limit.push_back(Lk); ftr_addr += Uk - Lk + 1; }
        c.cells += (uint64_t)(Uk - Lk + 1);
        const int32_t w1 = U[c1] - L[c1], w2 = U[c2] - L[c2];
        for (int32_t i = Lk; i < Uk + 1; ++i) {
            int8_t ptr = 0;
            bool Iptr = false, Dptr = false;
            const int32_t j = k - i;
            float match = -inf, insOp = -inf, delOp = -inf, insExt = -inf, delExt = -inf;
            const int32_t offset = i - Lk, offsetDiag = Lk - L[c2] + offset - 1, offsetUp = Lk - L[c1] + offset, offsetLeft = offsetUp - 1;
            const bool diag_ok = (offsetDiag >= 0) && (offsetDiag <= w2);
            const bool edge0 = (tile == 0) && (i == 0 || j == 0);
            if (k == 0 || diag_ok || edge0) {
                const float sim = columnScore(p, c.reference[(size_t)(reference_idx + j)].data(), c.query[(size_t)(query_idx + i)].data(), denominator);
                if (edge0) {
                    if (i == 0 && j == 0) match = sim;
                    else match = sim + p->gap_open + p->gap_extend * (float)std::max(0, std::max(reference_idx + j, query_idx + i) - 1);
                } else if (offsetDiag < 0) match = sim;
                else match = S[c2][offsetDiag] + sim;
            }
            const float gop_ref = c.gapOp[0][(size_t)(reference_idx + j)], gop_qry = c.gapOp[1][(size_t)(query_idx + i)];
            const float gex_ref = c.gapEx[0][(size_t)(reference_idx + j)], gex_qry = c.gapEx[1][(size_t)(query_idx + i)];
            if (offsetUp >= 0 && offsetUp <= w1) { delOp = S[c1][offsetUp] + gop_ref; delExt = D[b1][offsetUp] + gex_ref; }
            if (offsetLeft >= 0 && offsetLeft <= w1) { insOp = S[c1][offsetLeft] + gop_qry; insExt = I[b1][offsetLeft] + gex_qry; }
            I[b0][offset] = insOp; D[b0][offset] = delOp;
            if (insExt >= insOp) { I[b0][offset] = insExt; Iptr = true; }
            if (delExt >= delOp) { D[b0][offset] = delExt; Dptr = true; }
            if (match >= I[b0][offset]) {
                if (match >= D[b0][offset]) { S[c0][offset] = match; ptr = 0; }
                else { S[c0][offset] = D[b0][offset]; ptr = 2; }
            } else if (I[b0][offset] > D[b0][offset]) { S[c0][offset] = I[b0][offset]; ptr = 1; }
            else { S[c0][offset] = D[b0][offset]; ptr = 2; }
            if (S[c0][offset] < max_score - p->xdrop) S[c0][offset] = -inf;
            if (max_score_prime < S[c0][offset]) max_score_prime = S[c0][offset];
            if (k == marker - 1) CS[c0][offset] = (3 << 16) | (i & 0xFFFF);
            else if (k == marker) { CS[c0][offset] = (i & 0xFFFF); CI[b0][offset] = (1 << 16) | (i & 0xFFFF); CD[b0][offset] = (2 << 16) | (i & 0xFFFF); }
            else if (k >= marker + 1) {
                if (Iptr) CI[b0][offset] = (offsetLeft >= 0) ? CI[b1][offsetLeft] : kIB;
                else CI[b0][offset] = (offsetLeft >= 0 && CS[c1][offsetLeft] != -1) ? CS[c1][offsetLeft] : kIB;
                if (Dptr) CD[b0][offset] = (offsetUp >= 0) ? CD[b1][offsetUp] : kDB;
                else CD[b0][offset] = (offsetUp >= 0 && CS[c1][offsetUp] != -1) ? CS[c1][offsetUp] : kDB;
                if (ptr == 0) CS[c0][offset] = diag_ok ? CS[c2][offsetDiag] : -1;      // (the checker's definition of the unguarded read, talco_oracle.c)
                else if (ptr == 1) CS[c0][offset] = CI[b0][offset];
                else CS[c0][offset] = CD[b0][offset];
            }
            if (Iptr) ptr |= 0x04;
            if (Dptr) ptr |= 0x08;
            if (k <= marker) tb.push_back(ptr);
        }
        int32_t newL = Lk, newU = Uk;
        while (newL <= Uk && S[c0][newL - Lk] <= -inf) newL++;
        while (newU >= Lk && S[c0][newU - Lk] <= -inf) newU--;
        if (!converged && k < reference_length + query_length - 2) {
            const int32_t conv_I = uniformOrMinus1(CI[b0], newL - Lk, newU - newL), conv_D = uniformOrMinus1(CD[b0], newL - Lk, newU - newL);
            const int32_t conv_S = uniformOrMinus1(CS[c0], newL - Lk, newU - newL);
            if (conv_I == conv_D && conv_I == conv_S && prev_conv_s == conv_S && conv_I != -1) { converged = true; conv_value = prev_conv_s; conv_score = max_score_prime; }
            prev_conv_s = conv_S;
        }
        L[c2] = std::max(newL, std::max(0, k + 2 - reference_length));
        U[c2] = std::min(query_length - 1, newU + 1);
        max_score = (max_score_prime < 0) ? 0 : max_score_prime;
        last_k = k;
        if (converged && max_score > conv_score) { conv_logic = true; break; }
    }
    const int32_t n_ftr = (int32_t)ftr_length.size();
    int bad = 0;
    if (conv_logic) { conv_query_idx = conv_value & 0xFFFF; tb_state = (conv_value >> 16) & 0xFFFF; }
    else if (last_k < marker) { conv_query_idx = query_length - 1; conv_ref_idx = reference_length - 1; tb_start_addr = ftr_addr - 1; tb_start_ftr = last_k; tb_state = 0; last_tile = true; }
    else { conv_query_idx = CS[last_k % 3][0] & 0xFFFF; tb_state = (CS[last_k % 3][0] >> 16) & 0xFFFF; }
    if (conv_logic || last_k >= marker) {
        if (tb_state > 3 || n_ftr < 2) bad = 2;
        else {
            conv_ref_idx = marker - conv_query_idx - ((tb_state == 3) ? 1 : 0);
            tb_start_addr = ftr_addr - ftr_length[n_ftr - 1];
            tb_start_addr = (tb_state == 3) ? tb_start_addr - ftr_length[n_ftr - 2] + (conv_query_idx - ftr_lower_limit[n_ftr - 2])
                                            : tb_start_addr + (conv_query_idx - ftr_lower_limit[n_ftr - 1]);
            tb_start_ftr = (tb_state == 3) ? n_ftr - 2 : n_ftr - 1;
            if (tb_start_addr < 0 || (size_t)tb_start_addr >= tb.size()) bad = 3;
            else if (conv_ref_idx < 0) bad = 4;
        }
    }
    if (bad) { errorType = 3; last_tile = true; aln.clear(); freeMem(); return; }
    reference_idx += conv_ref_idx;
    query_idx += conv_query_idx;
    reference_length = (int32_t)c.reference.size() - reference_idx;
    query_length = (int32_t)c.query.size() - query_idx;
    if (reference_length < 0 || query_length < 0) { errorType = 3; aln.clear(); freeMem(); return; }
    if (reference_idx == (int32_t)c.reference.size() - 1 && query_idx < (int32_t)c.query.size() - 1) {
        for (int32_t q = 0; q < (int32_t)c.query.size() - query_idx - 1; ++q) aln.push_back(1);
        last_tile = true;
    }
    if (query_idx == (int32_t)c.query.size() - 1 && reference_idx < (int32_t)c.reference.size() - 1) {
        for (int32_t r = 0; r < (int32_t)c.reference.size() - reference_idx - 1; ++r) aln.push_back(2);
        last_tile = true;
    }
    if (reference_idx == (int32_t)c.reference.size() - 1 && query_idx == (int32_t)c.query.size() - 1) last_tile = true;
    tracebackF(ftr_length, ftr_lower_limit, tb_start_addr, tb_start_ftr, (int8_t)(tb_state % 3), conv_query_idx, conv_ref_idx, tb, aln, tile == 0);
    freeMem();
}

}  // namespace

extern "C" int twlf_align_batch(const twlo_params *p, int32_t n_pairs, int32_t seq_len, const float *freq, const float *gap_open, const float *gap_extend,
                                const int32_t *len, const int32_t *num, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out, int32_t threads,
                                uint64_t *cells_out)
{
    const size_t P = (size_t)p->P;
    uint64_t total = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : total)
    for (int32_t n = 0; n < n_pairs; ++n) {
        const int32_t R = len[2 * n], Q = len[2 * n + 1];
        aln_len_out[n] = 0; err_out[n] = 0;
        if (R <= 0 || Q <= 0) continue;
        // the per-pair containers, filled as alignment-cpu.cpp:70-86 fills them
        std::vector<std::vector<float>> freqRef((size_t)R, std::vector<float>(P, 0.0f)), freqQry((size_t)Q, std::vector<float>(P, 0.0f));
        std::vector<std::vector<float>> gapOp(2), gapEx(2);
        const float *fr = freq + ((size_t)n * 2 + 0) * (size_t)seq_len * P, *fq = freq + ((size_t)n * 2 + 1) * (size_t)seq_len * P;
        for (int32_t s = 0; s < R; ++s) for (size_t t = 0; t < P; ++t) freqRef[(size_t)s][t] = fr[P * (size_t)s + t];
        for (int32_t s = 0; s < Q; ++s) for (size_t t = 0; t < P; ++t) freqQry[(size_t)s][t] = fq[P * (size_t)s + t];
        const float *go = gap_open + (size_t)n * 2 * (size_t)seq_len, *ge = gap_extend + (size_t)n * 2 * (size_t)seq_len;
        for (int32_t s = 0; s < R; ++s) { gapOp[0].push_back(go[s]); gapEx[0].push_back(ge[s]); }
        for (int32_t s = 0; s < Q; ++s) { gapOp[1].push_back(go[seq_len + s]); gapEx[1].push_back(ge[seq_len + s]); }
        Faithful c{p, freqRef, freqQry, gapOp, gapEx, (float)num[2 * n], (float)num[2 * n + 1]};
        // Align_freq, :62-108
        std::vector<int8_t> aln;
        int32_t reference_idx = 0, query_idx = 0;
        bool last_tile = false, failed = false;
        int tile = 0;
        int16_t errorType = 0;
        while (!last_tile) {
            std::vector<int8_t> tile_aln;
            tileF(c, reference_idx, query_idx, tile_aln, last_tile, tile, errorType);
            if (tile_aln.empty()) { aln.clear(); failed = true; break; }
            for (int i = (int)tile_aln.size() - 1; i >= 0; --i) {
                if (i == (int)tile_aln.size() - 1 && tile > 0) continue;
                aln.push_back(tile_aln[(size_t)i]);
            }
            tile++;
        }
        total += c.cells;
        err_out[n] = errorType;
        if (!failed && (int64_t)aln.size() <= 2 * (int64_t)seq_len) {
            aln_len_out[n] = (int32_t)aln.size();
            memcpy(aln_out + (size_t)n * 2 * (size_t)seq_len, aln.data(), aln.size());
        }
    }
    if (cells_out) *cells_out = total;
    return 0;
}

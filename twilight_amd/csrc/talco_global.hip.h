// twilight_amd/csrc/talco_global.hip.h -- the LAST stage of the re-run chain: TALCO-XDrop with its DP rows in global memory, for bands of ANY width.
//
// What it computes: Talco_xdrop::Align_freq / Tile / Traceback (/root/reference/src/TALCO-XDrop.cpp:62-108, :233-689, :134-231), fp32, x86 TALCO_SIMD
// operation order, IEEE division -- the same results as every other kernel of this library, for pairs whose anti-diagonal band outgrew the 4608-row
// window of the widest register kernel.  Such bands only exist when fLen exceeds 4608, i.e. in the retries of the deferred pass, which raise fLen up to
// min(R, Q) (/root/reference/src/alignment-cpu.cpp:116-129); until round 5 they ended the run with TWL_ERR_UNSUPPORTED.
//
// How: the reference's own layout, with the loop over the cells of an anti-diagonal (:353) spread over the 1024 threads of ONE workgroup per pair.
// The rotating rows S[3], I[2], D[2], CS[3], CI[2], CD[2] (:277-311) live in global scratch, addressed by offset i - L exactly as the reference addresses
// them, so a read just outside a stored band (:535, :541) returns what the reference's arrays hold there.  Rows of diagonal k are only written on diagonal k
// and only read on k+1 / k+2, so the cells of a diagonal are independent; two workgroup barriers per diagonal (cells -> band ends and maximum -> convergence
// test).  Traceback pointers are bytes at [k][i - L(k)] with a fixed pitch.  This is a slow path by design (a few microseconds per anti-diagonal whatever
// the band): the fast paths are talco_nuc.hip.h; a pair only comes here after every register window has been outgrown.
#pragma once
#include "talco_kernel.hip.h"

namespace twl {

struct GArgs {
    KArgs k;                  // cols, len, num, outputs, queue / items, scoring (k.tb: this kernel's scratch, k.tb_words 32-bit words per workgroup)
    int32_t rowcap;           // elements per DP row: min(fLen, seq_len) + 2
};

template <int P>
__device__ __forceinline__ float global_column_score(const float *r, const float *q, const float *M, float gc, float denom)
{
    float numer = 0.0f;
    if constexpr (P == 6) {      // :378-395: per reference letter l the five products summed left to right, accumulated over l; then the gap-letter terms
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            float t[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) t[m] = (q[m] * M[5 * l + m]) * r[l];
            const float sl = (((t[0] + t[1]) + t[2]) + t[3]) + t[4];
            numer = (l == 0) ? sl : numer + sl;
        }
#pragma unroll
        for (int l = 0; l < 5; ++l) numer += (r[l] * q[5]) * gc;          // :394
#pragma unroll
        for (int m = 0; m < 5; ++m) numer += (r[5] * q[m]) * gc;          // :395
    } else {                     // :409-433: per letter the tail m = 16..20 first, then the eight pair sums left to right
        for (int l = 0; l < 21; ++l) {
            const float *row = M + 21 * l;
            const float rl = r[l];
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (q[t] * row[t]) * rl + (q[8 + t] * row[8 + t]) * rl;
#pragma unroll
            for (int m = 16; m < 21; ++m) numer += (rl * q[m]) * row[m];
            numer += ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
        }
        for (int l = 0; l < 21; ++l) numer += (r[l] * q[21]) * gc;        // :432
        for (int m = 0; m < 21; ++m) numer += (r[21] * q[m]) * gc;        // :433
    }
    return numer / denom;                                                  // :444 (correctly rounded fp32 division)
}

template <int P>
__global__ __launch_bounds__(1024, 1) void talco_global_kernel(GArgs ga)
{
    const KArgs &a = ga.k;
    constexpr int CW = P + 2, MS = P - 1, T = 1024;
    __shared__ float s_M[MS * MS];
    __shared__ int s_flow[kMaxMarker + 2];            // L of diagonals 0 .. marker (traceback addressing)
    __shared__ int s_red[3][4];                       // per diagonal mod 3 (reset two diagonals ahead of their use, before a barrier that every later poster has passed): {key of the maximum, first unpruned row, last unpruned row, -}
    __shared__ int s_conv[4];                         // {CS differs, CI or CD differs, -, -}
    __shared__ int s_misc[8];
    __shared__ int8_t s_rev[2 * kMaxMarker + 16];
    const int tid = threadIdx.x;
    for (int t = tid; t < MS * MS; t += T) s_M[t] = a.M[t];
    const float inf = (float)(2.0 * (double)a.xdrop + 1.0);     // :252
    const float xdropf = (float)a.xdrop;
    const size_t rowcap = (size_t)ga.rowcap;
    // scratch of this workgroup: 7 float rows, 7 int rows, then the traceback bytes [marker + 2][rowcap]
    uint32_t *base = a.tb + (size_t)blockIdx.x * (size_t)a.tb_words;
    float *fS[3], *fI[2], *fD[2];
    int *cS[3], *cI[2], *cD[2];
    for (int s = 0; s < 3; ++s) { fS[s] = reinterpret_cast<float *>(base) + rowcap * s; cS[s] = reinterpret_cast<int *>(base) + rowcap * (7 + s); }
    for (int s = 0; s < 2; ++s) {
        fI[s] = reinterpret_cast<float *>(base) + rowcap * (3 + s); fD[s] = reinterpret_cast<float *>(base) + rowcap * (5 + s);
        cI[s] = reinterpret_cast<int *>(base) + rowcap * (10 + s); cD[s] = reinterpret_cast<int *>(base) + rowcap * (12 + s);
    }
    int8_t *tb = reinterpret_cast<int8_t *>(base + 14 * rowcap);

    for (;;) {
        if (tid == 0) s_misc[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int item = s_misc[0];
        __syncthreads();
        if (item >= a.n_items) break;
        const int pair = a.items[item];
        const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
        const float gc = (a.gc_zero && a.gc_zero[pair]) ? 0.0f : a.gap_char;
        const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];      // :255, :269
        const float *colsR = a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * CW;
        const float *colsQ = a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * CW;
        int8_t *out = a.aln + (size_t)pair * 2 * (size_t)a.seq_len;
        const int marker = a.marker;
        int ref_idx = 0, qry_idx = 0, tile = 0, pos = 0, err = 0;
        bool last_tile = (R <= 0 || Q <= 0);
        unsigned long long cells = 0;
        long long steps_left = (long long)(R + Q + 2) * ((R + Q) / (max(marker, 2) - 1) + 4) + a.step_slack;

        while (!last_tile) {      // ---- Align_freq tile loop, :77-106 ----
            const int refLen = R - ref_idx, qLen = Q - qry_idx;
            if (refLen < 0 || qLen < 0 || steps_left < 0) { err = 3; break; }                 // :313-320
            const int fLen = min(a.flen, min(refLen, qLen));                                   // :258
            const size_t rowlen = (size_t)max(fLen, 1) + 1;
            if (rowlen > rowcap) { err = 3; break; }                                         // (the host sizes rowcap from fLen and seq_len: never)
            for (size_t t = tid; t < rowlen; t += T) {                                       // :277-311
#pragma unroll
                for (int s = 0; s < 3; ++s) { fS[s][t] = -1.0f; cS[s][t] = -1; }
#pragma unroll
                for (int s = 0; s < 2; ++s) { fI[s][t] = -1.0f; fD[s][t] = -1.0f; cI[s][t] = kIB; cD[s][t] = kDB; }
            }
            if (tid < 12) s_red[tid >> 2][tid & 3] = ((tid & 3) == 1) ? 0x7fffffff : (int)0x80000000;
            if (tid == 0) { s_conv[0] = 0; s_conv[1] = 0; }
            __syncthreads();
            int L0 = 0, U0 = 0, L1 = 1, U1 = -1, L2 = 2, U2 = -2;      // bands of diagonals k, k-1, k-2 (:296-297)
            float max_score = 0.0f, msp = -inf, conv_score = 0.0f;       // :259
            bool converged = false, conv_logic = false;
            int conv_value = 0, prev_conv_s = -1, last_k = 0, tile_err = 0;
            unsigned long long tcells = 0;      // (a tile that does not converge runs R + Q diagonals of a band this kernel exists for: > 2^32 cells)
            const int kEnd = refLen + qLen - 1;
            for (int k = 0; k < kEnd; ++k) {                             // :321
                const int c0 = k % 3, c1 = (k + 2) % 3, c2 = (k + 1) % 3, b0 = k & 1, b1 = (k + 1) & 1;
                if (L0 >= U0 + 1) { tile_err = 1; break; }               // :323-329
                if (U0 - L0 + 1 > fLen) { tile_err = 2; break; }         // :331-338
                if (k <= marker && tid == 0) s_flow[k] = L0;             // :340-344
                tcells += (unsigned long long)(unsigned)(U0 - L0 + 1);
                const int w1 = U1 - L1, w2 = U2 - L2;
                const float thr = max_score - xdropf;
                float lmax = -inf;
                int lmin = 0x7fffffff, lhi = (int)0x80000000;
                int *red = s_red[c0];
                // (the slots of diagonal k + 1, last read before the barrier of diagonal k - 1, are reset here: in front of this diagonal's barrier, behind which k + 1 posts)
                if (tid < 4) s_red[c2][tid] = (tid == 1) ? 0x7fffffff : (int)0x80000000;
                for (int i = L0 + tid; i <= U0; i += T) {                // :353
                    const int j = k - i;
                    const int offset = i - L0, offsetDiag = L0 - L2 + offset - 1, offsetUp = L0 - L1 + offset, offsetLeft = offsetUp - 1;      // :365-368
                    const bool diag_ok = offsetDiag >= 0 && offsetDiag <= w2;
                    const bool edge0 = (tile == 0) && (i == 0 || j == 0);
                    const float *rc = colsR + (size_t)(ref_idx + j) * CW, *qc = colsQ + (size_t)(qry_idx + i) * CW;
                    float match = -inf, insOp = -inf, delOp = -inf, insExt = -inf, delExt = -inf;
                    if (k == 0 || diag_ok || edge0) {                    // :369-371
                        float rv[P], qv[P];
#pragma unroll
                        for (int t = 0; t < P; ++t) { rv[t] = rc[t]; qv[t] = qc[t]; }
                        const float sim = global_column_score<P>(rv, qv, s_M, gc, denom);
                        if (edge0) {                                     // :445-448
                            if (i == 0 && j == 0) match = sim;
                            else { int far = max(ref_idx + j, qry_idx + i) - 1; far = far < 0 ? 0 : far; match = (sim + a.gap_open) + a.gap_extend * (float)far; }
                        } else if (offsetDiag < 0) match = sim;          // :449
                        else match = fS[c2][offsetDiag] + sim;           // :450
                    }
                    if (offsetUp >= 0 && offsetUp <= w1) { delOp = fS[c1][offsetUp] + rc[P]; delExt = fD[b1][offsetUp] + rc[P + 1]; }          // :456-459
                    if (offsetLeft >= 0 && offsetLeft <= w1) { insOp = fS[c1][offsetLeft] + qc[P]; insExt = fI[b1][offsetLeft] + qc[P + 1]; }  // :460-463
                    float Iv = insOp, Dv = delOp;                        // :464-475
                    bool Iptr = false, Dptr = false;
                    if (insExt >= insOp) { Iv = insExt; Iptr = true; }
                    if (delExt >= delOp) { Dv = delExt; Dptr = true; }
                    float Sv; int ptr;                                   // :477-494
                    if (match >= Iv) { if (match >= Dv) { Sv = match; ptr = 0; } else { Sv = Dv; ptr = 2; } }
                    else if (Iv > Dv) { Sv = Iv; ptr = 1; }
                    else { Sv = Dv; ptr = 2; }
                    if (Sv < thr) Sv = -inf;                             // :495-497
                    fI[b0][offset] = Iv; fD[b0][offset] = Dv; fS[c0][offset] = Sv;
                    if (Sv > lmax) lmax = Sv;                            // :501-503
                    if (Sv > -inf) { lmin = min(lmin, i); lhi = max(lhi, i); }
                    if (k == marker - 1) cS[c0][offset] = (3 << 16) | (i & 0xFFFF);            // :520-526
                    else if (k == marker) { cS[c0][offset] = i & 0xFFFF; cI[b0][offset] = (1 << 16) | (i & 0xFFFF); cD[b0][offset] = (2 << 16) | (i & 0xFFFF); }
                    else if (k >= marker + 1) {                          // :527-547
                        int CIn, CDn;
                        if (Iptr) CIn = (offsetLeft >= 0) ? cI[b1][offsetLeft] : kIB;
                        else { const int v = (offsetLeft >= 0) ? cS[c1][offsetLeft] : -1; CIn = (v != -1) ? v : kIB; }
                        if (Dptr) CDn = (offsetUp >= 0) ? cD[b1][offsetUp] : kDB;
                        else { const int v = (offsetUp >= 0) ? cS[c1][offsetUp] : -1; CDn = (v != -1) ? v : kDB; }
                        cI[b0][offset] = CIn; cD[b0][offset] = CDn;
                        // (M without a diagonal predecessor: "unset", as the checker and every other kernel define the reference's out-of-range read, :541)
                        cS[c0][offset] = (ptr == 0) ? (diag_ok ? cS[c2][offsetDiag] : -1) : ((ptr == 1) ? CIn : CDn);
                    }
                    if (k <= marker) tb[(size_t)k * rowcap + (size_t)offset] = (int8_t)(ptr | (Iptr ? 4 : 0) | (Dptr ? 8 : 0));                // :548-557
                }
                // ---- the diagonal's maximum and its first / last unpruned row (:501-503, :563-583) ----
                if (lmax > -inf) atomicMax(&red[0], f2key(lmax));
                if (lmin != 0x7fffffff) { atomicMin(&red[1], lmin); atomicMax(&red[2], lhi); }
                __syncthreads();
                {
                    const int key = red[0];
                    if (key != (int)0x80000000) { const float g = key2f(key); if (msp < g) msp = g; }
                }
                int newL = red[1], newU = red[2];
                if (newL == 0x7fffffff) { newL = U0 + 1; newU = L0 - 1; }      // every cell pruned: where the reference's two scans stop
                if (!converged && k < kEnd - 1 && k >= marker - 1) {            // :585-595 (before the marker the pointer rows hold -1 / -2 / -3: conv_S = -1, never converged)
                    int conv_S = -1;
                    bool all3 = false;
                    if (newL <= newU) {
                        const int vS = cS[c0][newL - L0];
                        bool badS = false, badID = false;
                        for (int i = newL + tid; i <= newU; i += T) {
                            badS |= cS[c0][i - L0] != vS;
                            badID |= (cI[b0][i - L0] != vS) | (cD[b0][i - L0] != vS);
                        }
                        if (badS) atomicOr(&s_conv[0], 1);
                        if (badID) atomicOr(&s_conv[1], 1);
                        __syncthreads();
                        const bool anyS = s_conv[0] != 0, anyID = s_conv[1] != 0;
                        __syncthreads();
                        if (tid == 0) { s_conv[0] = 0; s_conv[1] = 0; }
                        if (!anyS) { conv_S = vS; all3 = !anyID; }
                    }
                    // (an empty surviving band: conv_S = -1, as every kernel of this library and the checker's tested cases have it)
                    if (all3 && prev_conv_s == conv_S && conv_S != -1) { converged = true; conv_value = prev_conv_s; conv_score = msp; }
                    prev_conv_s = conv_S;
                }
                {                                                         // :597-604
                    const int nL = max(max(newL, k + 2 - refLen), 0), nU = min(newU + 1, qLen - 1);
                    L2 = L1; U2 = U1; L1 = L0; U1 = U0; L0 = nL; U0 = nU;
                }
                max_score = (msp < 0.0f) ? 0.0f : msp;                   // :607
                last_k = k;
                if (converged && max_score > conv_score) { conv_logic = true; break; }      // :609-612
            }
            __syncthreads();
            steps_left -= (long long)(last_k + 1);
            cells += tcells;
            if (tile_err) { err = tile_err; break; }
            // ---- tile exit, :615-682 ----
            int conv_q = 0, conv_r = 0, tb_state = 0, start_k = 0;
            bool bad = false;
            if (!conv_logic && last_k >= marker) conv_value = cS[last_k % 3][0];               // :633-635
            if (conv_logic || last_k >= marker) {
                conv_q = conv_value & 0xFFFF;
                tb_state = (conv_value >> 16) & 0xFFFF;
                if (tb_state > 3) bad = true;          // boundary sentinel / unset: the reference indexes out of range here (errorType 3, as the checker defines it)
                else { conv_r = marker - conv_q - ((tb_state == 3) ? 1 : 0); start_k = (tb_state == 3) ? marker - 1 : marker; if (conv_r < 0) bad = true; }
            } else { conv_q = qLen - 1; conv_r = refLen - 1; start_k = last_k; tb_state = 0; last_tile = true; }      // :625-632
            if (bad) { err = 3; break; }
            ref_idx += conv_r; qry_idx += conv_q;                        // :654-655
            if (R - ref_idx < 0 || Q - qry_idx < 0) { err = 3; break; }  // :659-668
            int tailDir = 0, tailLen = 0;
            if (ref_idx == R - 1 && qry_idx < Q - 1) { tailDir = 1; tailLen = Q - qry_idx - 1; last_tile = true; }      // :671-674
            if (qry_idx == Q - 1 && ref_idx < R - 1) { tailDir = 2; tailLen = R - ref_idx - 1; last_tile = true; }      // :675-678
            if (ref_idx == R - 1 && qry_idx == Q - 1) last_tile = true;                                                 // :679
            // ---- Traceback, :134-231: one thread walks the pointer bytes ----
            if (tid == 0) {
                int n = 0, kk = start_k, ii = conv_q, qi = conv_q, ri = conv_r, st = tb_state % 3;
                const bool first = (tile == 0);
                while (kk >= 0) {
                    const int off = ii - s_flow[kk];
                    const int v = (off >= 0 && (size_t)off < rowcap) ? (int)tb[(size_t)kk * rowcap + (size_t)off] : 0;
                    int dir;
                    if (st == 0) {
                        st = v & 3;
                        if (st == 0) dir = 0;
                        else if (st == 1) { dir = 1; st = (v & 4) ? 1 : 0; }
                        else { dir = 2; st = (v & 8) ? 2 : 0; }
                    } else if (st == 1) { dir = 1; st = (v & 4) ? 1 : 0; }
                    else { dir = 2; st = (v & 8) ? 2 : 0; }
                    if (dir == 0) { kk -= 2; ii -= 1; qi--; ri--; }
                    else if (dir == 1) { kk -= 1; ii -= 1; qi--; }
                    else { kk -= 1; ri--; }
                    if (n < (int)sizeof(s_rev)) s_rev[n] = (int8_t)dir;
                    ++n;
                    if (first && (ri < 0 || qi < 0)) break;
                    if (ii < 0) break;                 // defensive: a pointer chain left the tile (never on valid data)
                }
                if (first) {
                    while (ri > -1) { if (n < (int)sizeof(s_rev)) s_rev[n] = 2; ++n; ri--; }
                    while (qi > -1) { if (n < (int)sizeof(s_rev)) s_rev[n] = 1; ++n; qi--; }
                }
                s_misc[1] = n;
            }
            __syncthreads();
            const int n = s_misc[1];
            const int skip = (tile > 0) ? 1 : 0;                         // :98-102
            const int cnt = n - skip;
            if (n > (int)sizeof(s_rev) || pos + cnt + tailLen > 2 * a.seq_len) { err = 3; break; }
            for (int t = tid; t < cnt; t += T) out[pos + t] = s_rev[n - 1 - skip - t];
            for (int t = tid; t < tailLen; t += T) out[pos + cnt + t] = (int8_t)tailDir;
            pos += cnt + tailLen;
            tile += 1;
            __syncthreads();
        }
        __syncthreads();
        if (tid == 0) {
            a.err[pair] = (int16_t)err;
            a.aln_len[pair] = (err == 0) ? pos : 0;
            a.cells[pair] = cells;
        }
        __syncthreads();
    }
}

}  // namespace twl

// twilight_amd/csrc/talco_i16.hip.h -- leaf x leaf pairs as a 16-bit PACKED recurrence: two query rows per 32-bit lane register (round 5).
//
// What it computes: Talco_xdrop::Align_freq / Tile / Traceback (/root/reference/src/TALCO-XDrop.cpp:62-108, :233-689, :134-231) for pairs of SINGLE SEQUENCES,
// bit for bit what talco_lean_kernel computes for them in fp32.
//
// Why it is exact.  With one sequence on either side every profile column is one-hot with the value 1.0 (alignment-helper.cpp:27), so every product
// (q[m]*M[l][m])*r[l] of the column score (:378-395) is M[l*][m*] or +-0, the denominator is 1 (:269), the gap penalties are the two parameters themselves
// (calculatePSGP, alignment-helper.cpp:188-204: no gap counts): every operand of the recurrence (:445-497) is a small integer and every fp32 result the integer
// result.  A tile restarts at 0 (:77-93), so its scores stay below 18 * (k / 2 + 1); "-inf" is -(2 * xdrop + 1) (:252) and nothing falls below -inf + gapOpen +
// gapExtend (a state with no valid neighbour IS -inf, :468-475).  The host checks the predicate (integral matrix / gap values of small magnitude, X-drop <= 16 000,
// no ambiguity letter in either sequence, rows < 32 000); the kernel ends a tile that runs past 3 400 anti-diagonals, a denominator other than 1, a gap letter or an
// ambiguity letter with the internal re-run code (kErrGuard): the pair then runs on the fp32 kernels.
//
// Why bother.  The diagonal step is bound by the NUMBER of instructions a SIMD issues; v_pk_add_i16 / v_pk_max_i16 / v_pk_sub_i16 (clamp) /
// v_pk_ashrrev_i16 advance two cells per instruction, v_bfi_b32 selects two cells by a packed mask, and a compare is a saturating subtraction whose sign is smeared
// over the half-word.  Rows (2l, 2l + 1) share lane l of a 128-row sub-block; a wave owns R2 consecutive sub-blocks, so the per-wave part of a diagonal (barrier,
// band update: ~55 instructions) is paid once per 128 R2 rows.
//   * column score: the reference ring holds ONE selector word per staged column ({code | 0x0C00, (4 + code) | 0x0C00}); a row pair keeps two byte tables
//     {M[l][m*] + bias, l = A, C, G, T}; v_perm_b32 picks both cells' scores, one packed subtraction takes the bias off;
//   * row i - 1: the upper row of a lane has it in the same register; the lower row takes the lane below's upper row: one DPP wave_shr + one v_alignbit per value;
//   * band tests: "row in band(k)" by two packed subtractions; "in band(k-1)" is last diagonal's mask, "row i-1 in band(k-1)" that mask moved up a row the way the
//     values move, "row i-1 in band(k-2)" last diagonal's moved mask;
//   * convergence pointers (:520-547) as 16-bit words (state << 14 | row: a pointer names a row of the marker diagonals, <= 1024; -1 / -2 / -3 stay themselves).
// Throughput launches only (pairs from a queue); the tile-parallel remainder of a leaf level stays on the fp32 kernels.
#pragma once
#include "talco_nuc.hip.h"

namespace twl {

typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i16x2 as_s2(unsigned v) { return __builtin_bit_cast(i16x2, v); }
__device__ __forceinline__ unsigned as_u(i16x2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return as_u(as_s2(a) + as_s2(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return as_u(as_s2(a) - as_s2(b)); }
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) { return as_u(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
// 0xFFFF in a half where a < b (signed 16-bit), 0 elsewhere: the sign of the saturating difference, smeared
__device__ __forceinline__ unsigned pk_lt(unsigned a, unsigned b) { return as_u(__builtin_elementwise_sub_sat(as_s2(a), as_s2(b)) >> (short)15); }
__device__ __forceinline__ unsigned bfi(unsigned m, unsigned a, unsigned b) { return (m & a) | (~m & b); }      // v_bfi_b32: a where the mask is set, b elsewhere
__device__ __forceinline__ unsigned both(int v) { return ((unsigned)v & 0xFFFFu) | ((unsigned)v << 16); }      // the same 16-bit value in both halves
__device__ __forceinline__ unsigned dpp_shr1_u(unsigned old, unsigned src) { return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned dpp_ror1_u(unsigned src) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)src, 0x13C, 0xf, 0xf, false); }
// the packed values of rows i - 1: upper half <- own lower half, lower half <- the lane below's upper half (lane 0: `below`'s upper half)
__device__ __forceinline__ unsigned rows_up(unsigned below, unsigned x) { return __builtin_amdgcn_alignbit(x, dpp_shr1_u(below, x), 16); }

template <int W, int R2>
struct I16Cfg {
    static constexpr int SBR = 128 * R2;          // rows of a wave's super block
    static constexpr int NSB = W * R2;            // 128-row sub-blocks resident at once
    static constexpr int WINDOW = 128 * NSB;      // rows
    static constexpr int CAP = WINDOW + 128;      // ring columns (two 64-column stages ahead)
    static constexpr int THREADS = 64 * W;
};
constexpr int kI16MaxDiag = 3400;                 // 18 * (3400 / 2 + 1) < 2^15: a tile that runs longer goes to the fp32 kernel

template <int W, int R2, int MINW>
__global__ __launch_bounds__(64 * W, MINW) void talco_i16_kernel(NArgs a)
{
    using C = I16Cfg<W, R2>;
    constexpr int SBR = C::SBR, NSB = C::NSB, WINDOW = C::WINDOW, CAP = C::CAP;
    constexpr int P = 6, F4 = 2;
    constexpr int I_UNSET = -1, I_IB = kIB, I_DB = kDB;      // the reference's -1 / I_BOUNDARY / D_BOUNDARY as 16-bit words

    __shared__ unsigned s_sel[CAP + 2];          // selector word of every staged reference column; [CAP] repeats [0]
    struct ParBuf {
        int cd[WINDOW + 4];          // offset-addressed mirror of the reference's CD rows (sign-extended 16-bit pointer words)
        uint4 exch[W];               // mailbox: lane 63 of a wave's last sub-block -> lane 0 of the next wave's first {S, I, CS, CI} (packed row pairs)
        int red[4];                  // {running max S, low-end tag, high-end tag, -}
        int conv[4];                 // {vmin, vmax, flags, -}
        int edge[2 * NSB];           // phase C: pointer word of the first / last unpruned row of every 128-row sub-block (the cheap pre-test)
        int4 trash[64];
    };
    __shared__ ParBuf s_par[2];
    __shared__ int s_misc[8];
    __shared__ int8_t s_rev[2 * kMaxMarker + 16];
    constexpr unsigned O_CD = (unsigned)offsetof(ParBuf, cd), O_EXCH = (unsigned)offsetof(ParBuf, exch), O_RED = (unsigned)offsetof(ParBuf, red),
                       O_CONV = (unsigned)offsetof(ParBuf, conv), O_TRASH = (unsigned)offsetof(ParBuf, trash), O_EDGE = (unsigned)offsetof(ParBuf, edge);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tb = a.tb + (size_t)blockIdx.x * (size_t)a.tb_words;
    // ---- the scoring of the launch as integers (the host checked that they are) ----
    const int NEG = -(2 * a.xdrop + 1);                         // TALCO-XDrop.cpp:252
    const unsigned NEG2 = both(NEG);
    const int go = (int)a.gap_open, ge = (int)a.gap_extend;
    const unsigned GOP2 = both(go), GEX2 = both(ge);
    int mmin = 0;
    for (int l = 0; l < 4; ++l) for (int m = 0; m < 4; ++m) mmin = min(mmin, (int)a.M[5 * l + m]);
    const int bias = -mmin;
    const unsigned BIAS2 = both(bias);

    for (;;) {
        if (threadIdx.x == 0) s_misc[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (item >= a.n_items) break;
        const int pair = __builtin_amdgcn_readfirstlane(a.items[item]);
        const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
        const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];
        const float4 *colsR = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * (P + 2));
        const float4 *colsQ = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * (P + 2));
        int ref_idx = 0, qry_idx = 0, tile = 0, pos = 0, err = 0;
        bool last_tile = (R <= 0 || Q <= 0);
        int8_t *out = a.aln + (size_t)pair * 2 * (size_t)a.seq_len;
        unsigned long long cells = 0;
        long long steps_left = (long long)(R + Q + 2) * ((R + Q) / (max(a.marker, 2) - 1) + 4) + a.step_slack;
        if (!last_tile && (denom != 1.0f || Q > 32000)) { err = kErrGuard; last_tile = true; }
        int dbg_lastk = 0, dbg_conv = 0, dbg_L = 0, dbg_U = 0;
        bool guardBad = false;
        const int marker = a.marker;

        while (!last_tile) {   // ---- Align_freq tile loop, TALCO-XDrop.cpp:77-106 ----
            const int refLen = R - ref_idx, qLen = Q - qry_idx;
            const int fLen = min(a.flen, min(refLen, qLen));                          // :258
            const int fcap = min(fLen, SBR * (W - 1));      // below it the band touches at most W super blocks and a block that leaves it is not needed again on the same diagonal
            // ---- the wave's rows: sub-block r = rows 128 (R2 sblk + r) + 2 lane + {0, 1}, every per-row quantity a packed pair {even row, odd row} ----
            unsigned S1[R2], I1[R2], D1[R2], LS2[R2];
            unsigned CS1[R2], CI1[R2], CD1[R2], LCS2[R2];
            unsigned TabE[R2], TabO[R2];                  // score bytes of the two rows against reference letter A, C, G, T (+ bias)
            unsigned pOut[R2], pLeftOut[R2];          // last diagonal's masks (0xFFFF = NOT): row in band(k-1); row i-1 in band(k-1)
            uint32_t tbA[R2], tbB[R2];
            int sblk;
            int fresh;
            unsigned raB;                             // byte address of the selector word of column k - (first odd row of the super block) ... see sel_addr

            auto load_q = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    unsigned tab[2];
                    bool bad = false;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = SBR * sblk + 128 * r + 2 * lane + h;
                        const bool ok = qry_idx + i < Q;
                        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
                        if (ok) { c0 = colsQ[F4 * (size_t)(qry_idx + i)]; c1 = colsQ[F4 * (size_t)(qry_idx + i) + 1]; }
                        // the row's letter m*: its byte table {M[l][m*] + bias}
                        const int m = (c0.x != 0.0f) ? 0 : ((c0.y != 0.0f) ? 1 : ((c0.z != 0.0f) ? 2 : 3));
                        const float one = (m == 0) ? c0.x : ((m == 1) ? c0.y : ((m == 2) ? c0.z : c0.w));
                        bad = bad | (ok && (one != 1.0f || c1.x != 0.0f || c1.y != 0.0f || (c0.x != 0.0f) + (c0.y != 0.0f) + (c0.z != 0.0f) + (c0.w != 0.0f) != 1));
                        unsigned t = 0;
#pragma unroll
                        for (int l = 0; l < 4; ++l) t |= (unsigned)((int)a.M[5 * l + m] + bias) << (8 * l);
                        tab[h] = t;
                    }
                    TabE[r] = tab[0]; TabO[r] = tab[1];
                    guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
                }
            };
            auto load_ring_block = [&](int B) __attribute__((always_inline)) {      // 64 reference columns -> their selector words
                const int col = 64 * B + lane;
                const int sl = col % CAP;
                float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
                const bool ok = col < refLen;
                if (ok) { c0 = colsR[F4 * (size_t)(ref_idx + col)]; c1 = colsR[F4 * (size_t)(ref_idx + col) + 1]; }
                const int code = (c0.x != 0.0f) ? 0 : ((c0.y != 0.0f) ? 1 : ((c0.z != 0.0f) ? 2 : 3));
                const float one = (code == 0) ? c0.x : ((code == 1) ? c0.y : ((code == 2) ? c0.z : c0.w));
                const bool bad = ok && (one != 1.0f || c1.x != 0.0f || c1.y != 0.0f || (c0.x != 0.0f) + (c0.y != 0.0f) + (c0.z != 0.0f) + (c0.w != 0.0f) != 1);
                const unsigned word = ((unsigned)code | 0x0C00u) | (((unsigned)(4 + code) | 0x0C00u) << 16);
                s_sel[sl] = word;
                if (sl == 0) s_sel[CAP] = word;
                guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
            };
            // column of the ODD row of sub-block 0, lane l, on diagonal k: j = k - (SBR sblk + 2l + 1); the even row's is j + 1.  Sub-block r: 128 r columns lower.
            auto sel_addr = [&](int k) __attribute__((always_inline)) {
                int rs = (k - SBR * sblk - 2 * lane - 1) % CAP;
                rs += (rs < 0) ? CAP : 0;
                raB = (unsigned)rs * 4u + lds_off(s_sel);
            };

            sblk = w;
            sel_addr(0);
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                S1[r] = I1[r] = D1[r] = LS2[r] = NEG2;
                CS1[r] = CI1[r] = CD1[r] = LCS2[r] = 0u;
                pOut[r] = pLeftOut[r] = 0xFFFFFFFFu;      // bands k-1 and k-2 are empty when the tile begins
                tbA[r] = tbB[r] = 0u;
            }
            fresh = 0;
            load_q();
            int hiBlk = 1;
            if (w == 0 % W) load_ring_block(0);
            if (w == 1 % W) load_ring_block(1);
            for (int t = threadIdx.x; t < WINDOW + 4; t += C::THREADS) { s_par[0].cd[t] = I_DB; s_par[1].cd[t] = I_DB; }      // :308
            if (threadIdx.x == 0) {
                s_par[0].red[0] = s_par[1].red[0] = NEG;
                s_par[0].red[1] = s_par[0].red[2] = s_par[1].red[1] = s_par[1].red[2] = 0;
                for (int t = 0; t < 2; ++t) { s_par[t].conv[0] = 0x7fffffff; s_par[t].conv[1] = (int)0x80000000; s_par[t].conv[2] = 0; }
                s_misc[4] = 0;
            }
            __syncthreads();

            // ---- Tile, TALCO-XDrop.cpp:233-689 ----
            int Lk = 0, Uk = 0, sLo1 = 0x3fffffff, sLo2p = 0x3fffffff;      // bands: k; k-1 as (low, width-1); k-2 as (low + 1, width-1); empty = unreachable low
            int sW1 = 0, sW2 = 0;                        // (an empty band: unreachable low, width-1 = 0: no row matches)
            int vwid1 = 0;                               // width of diagonal k-1 (0 when empty): the stale CD slot
            unsigned vcells = 0;
            int msp = NEG, convS = 0;                    // running maximum (:259), score at convergence (:594)
            bool converged = false;
            int conv_value = 0, prev_conv_s = I_UNSET;
            const int kEnd = refLen + qLen - 1;
            int k = 0;
            int tile_err = 0;
            bool go_on = true, conv_logic = false;
            bool spec = true;
            bool tbPending = false;
            unsigned tbOff = 0u;                         // byte offset of the current group of 8 diagonals
            unsigned vcur = lds_off(&s_par[0]), vprev = lds_off(&s_par[1]);
            const unsigned parx = lds_off(&s_par[0]) ^ lds_off(&s_par[1]);
            const unsigned relTrash = O_TRASH + (unsigned)lane * 16u;
            const unsigned mbRel = (lane == 63) ? O_EXCH + 16u * (unsigned)w : relTrash;
            const unsigned exRel = O_EXCH + 16u * (unsigned)((w + W - 1) % W);
            int tbL0 = 0, tbU0 = 0;                      // (traceback words only where the band was: see talco_lean_kernel)
            int tbMust = 0;

            // 0xFFFF in a half whose row lies OUTSIDE [lo, hi]
            auto out_mask = [&](int r, int lo, int hi, int shift) __attribute__((always_inline)) {
                const unsigned idx = both(SBR * sblk + 128 * r - shift) + (unsigned)(2 * lane) * 0x00010001u + 0x00010000u;      // {even row, odd row} (- shift)
                return as_u(as_s2(pk_sub(idx, both(lo)) | pk_sub(both(hi), idx)) >> (short)15);
            };

            auto step = [&](auto PHtag) __attribute__((always_inline)) {
                constexpr int PH = decltype(PHtag)::value;
                constexpr bool TB = (PH != 2), CONV = (PH != 0);
                const unsigned kk16 = (unsigned)(k + 1) << 16;
                const int width1 = Uk - Lk;
                vcells += (unsigned)(width1 + 1);
                const int thr = max(msp, 0) - a.xdrop;                           // :495 with :607
                const unsigned THR2 = both(thr);
                bool special = false;
                if (__builtin_expect(spec, 0)) {
                    special = (k == 0) | ((tile == 0) && (Lk == 0 || Uk == k));
                    spec = special;
                }
                int staleCD = I_DB;
                if constexpr (PH == 2) staleCD = lds_ld<int>(vprev + 4u * (unsigned)vwid1 + O_CD);
                const unsigned vTrashRed = vcur + relTrash;
                const int b = SBR * sblk;
                int sLowC = 0x7fffffff, sHighC = -1;
                // the super block takes part when the band touches it or will reach its first row on the next diagonal: Lk - (SBR - 1) <= b <= Uk + 1
                if ((unsigned)(b - (Lk - (SBR - 1))) <= (unsigned)(width1 + SBR)) {
                    TWL_SETPRIO(2);
                    const nuc_i4 e = lds_ld<nuc_i4>(vprev + exRel);      // lane 63 of the wave below: its packed {S, I, CS, CI}
                    unsigned belowS = (unsigned)e.x, belowI = (unsigned)e.y, belowCS = (unsigned)e.z, belowCI = (unsigned)e.w;
                    unsigned belowOut = ((unsigned)(b - 1 - sLo1) <= (unsigned)sW1) ? 0u : 0xFFFF0000u;      // row b-1 in band(k-1)?  (its mask, as the upper half of a lane below)
#pragma unroll
                    for (int r = 0; r < R2; ++r) {
                        const int b128 = b + 128 * r;
                        // (a sub-block the band does not touch: nothing to compute, but its registers feed the next one's lane 0)
                        const bool act = (unsigned)(b128 - (Lk - 127)) <= (unsigned)(width1 + 128);
                        unsigned upOut, leftOut, diagOut;
                        if (__builtin_expect(fresh != 0, 0)) {
                            upOut = out_mask(r, sLo1, sLo1 + sW1, 0);
                            leftOut = out_mask(r, sLo1, sLo1 + sW1, 1);
                            diagOut = out_mask(r, sLo2p - 1, sLo2p - 1 + sW2, 1);
                        } else {
                            upOut = pOut[r];
                            leftOut = rows_up(belowOut, pOut[r]);
                            diagOut = pLeftOut[r];
                        }
                        const unsigned nextBelowOut = dpp_ror1_u(pOut[r]);
                        const unsigned outK = out_mask(r, Lk, Uk, 0);
                        pOut[r] = outK; pLeftOut[r] = leftOut;
                        // ---- rows i - 1 ----
                        const unsigned LS1 = rows_up(belowS, S1[r]), LI1 = rows_up(belowI, I1[r]);
                        unsigned LCS1 = 0u, LCI1 = 0u;
                        if constexpr (CONV) { LCS1 = rows_up(belowCS, CS1[r]); LCI1 = rows_up(belowCI, CI1[r]); }
                        const unsigned nbS = dpp_ror1_u(S1[r]), nbI = dpp_ror1_u(I1[r]);
                        unsigned nbCS = 0u, nbCI = 0u;
                        if constexpr (CONV) { nbCS = dpp_ror1_u(CS1[r]); nbCI = dpp_ror1_u(CI1[r]); }
                        if (act) {
                            // ---- column score: one selector word per cell, v_perm_b32 out of the rows' byte tables ----
                            const unsigned aSel = raB - (unsigned)(128 * 4 * r) + ((raB < lds_off(s_sel) + (unsigned)(128 * 4 * r)) ? (unsigned)(CAP * 4) : 0u);
                            const unsigned wOdd = lds_ld<unsigned>(aSel), wEven = lds_ld<unsigned>(aSel + 4u);
                            const unsigned sel = bfi(0x0000FFFFu, wEven, wOdd);
                            const unsigned sim = pk_sub(__builtin_amdgcn_perm(TabO[r], TabE[r], sel), BIAS2);
                            // ---- recurrence, :445-497 ----
                            unsigned match = bfi(diagOut, NEG2, pk_add(LS2[r], sim));
                            if (__builtin_expect(special, 0)) {
                                int mv[2];
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const int i = b128 + 2 * lane + h, j = k - i;
                                    int m = (int)(short)(match >> (16 * h));
                                    const int sv = (int)(short)(sim >> (16 * h));
                                    if (k == 0) m = sv;
                                    else if (i == 0 || j == 0) { int far = max(i, j) - 1; far = far < 0 ? 0 : far; m = (sv + go) + ge * far; }
                                    mv[h] = m;
                                }
                                match = ((unsigned)mv[0] & 0xFFFFu) | ((unsigned)mv[1] << 16);
                            }
                            const unsigned delOp = pk_add(S1[r], GOP2), delExt = pk_add(D1[r], GEX2);      // :456-463
                            const unsigned insOp = pk_add(LS1, GOP2), insExt = pk_add(LI1, GEX2);
                            // an invalid neighbour makes the state -inf and "extend" wins the tie (:468-475)
                            const unsigned dLt = pk_lt(delExt, delOp), iLt = pk_lt(insExt, insOp);
                            const unsigned DptrM = bfi(dLt, upOut, 0xFFFFFFFFu), IptrM = bfi(iLt, leftOut, 0xFFFFFFFFu);
                            const unsigned Dv = bfi(upOut, NEG2, pk_max(delExt, delOp));
                            const unsigned Iv = bfi(leftOut, NEG2, pk_max(insExt, insOp));
                            const unsigned gI = pk_lt(Dv, Iv);                                       // :477-494: I > D (D wins ties)
                            const unsigned Gv = pk_max(Iv, Dv);
                            const unsigned notM = pk_lt(match, Gv);                                  // M wins when match >= G
                            const unsigned Sv0 = pk_max(match, Gv);
                            const unsigned dead = pk_lt(Sv0, THR2) | outK;                           // :495-497, and the band
                            const unsigned Sv = bfi(dead, NEG2, Sv0);
                            if constexpr (CONV) {                                                    // :520-547
                                const unsigned row2 = both(b128) + (unsigned)(2 * lane) * 0x00010001u + 0x00010000u;
                                unsigned CSn, CIn, CDn;
                                if (PH == 1 && k == marker - 1) { CSn = row2 | 0xC000C000u; CIn = CI1[r]; CDn = CD1[r]; }
                                else if (PH == 1) { CSn = row2; CIn = row2 | 0x40004000u; CDn = row2 | 0x80008000u; }
                                else {
                                    const unsigned lcsUnset = ~pk_lt(LCS1, both(I_UNSET)) & ~pk_lt(both(I_UNSET), LCS1);      // LCS1 == -1
                                    const unsigned viaS = bfi(lcsUnset, both(I_IB), LCS1);
                                    CIn = bfi(leftOut, both(I_IB), bfi(IptrM, LCI1, viaS));
                                    const unsigned cdUp = bfi(upOut, both(staleCD), CD1[r]);        // above the stored band the reference reads the stale slot (:535)
                                    const unsigned csUnset = ~pk_lt(CS1[r], both(I_UNSET)) & ~pk_lt(both(I_UNSET), CS1[r]);
                                    CDn = bfi(DptrM, cdUp, bfi(csUnset, both(I_DB), CS1[r]));
                                    const unsigned viaGap = bfi(gI, CIn, CDn);
                                    CSn = bfi(notM, viaGap, bfi(diagOut, both(I_UNSET), LCS2[r]));
                                }
                                CS1[r] = CSn; CI1[r] = CIn; CD1[r] = CDn;
                                if (PH == 2 || k == marker) {      // the CD mirror, rows in the band only
                                    const int iE = b128 + 2 * lane;
                                    const unsigned aE = ((outK & 0x0000FFFFu) == 0u) ? vcur + 4u * (unsigned)(iE - Lk) + O_CD : vTrashRed;
                                    const unsigned aO = ((outK & 0xFFFF0000u) == 0u) ? vcur + 4u * (unsigned)(iE + 1 - Lk) + O_CD : vTrashRed + 4u;
                                    lds_st<int>(aE, (int)(short)CDn);
                                    lds_st<int>(aO, (int)CDn >> 16);
                                }
                            }
                            S1[r] = Sv; I1[r] = Iv; D1[r] = Dv;
                            // ---- reductions of this diagonal (:501-503, :563-583) ----
                            {
                                const int sE = (int)(short)Sv, sO = (int)Sv >> 16;
                                const int smax = max(sE, sO);
                                const unsigned long long gm = __builtin_amdgcn_ballot_w64(smax > msp);
                                if (gm != 0ull) ds_max_i32_off<O_RED>(__builtin_amdgcn_inverse_ballot_w64(gm) ? vcur : vTrashRed, smax);
                            }
                            // first / last unpruned row of the sub-block: the end lanes of "some half alive", then which half of each
                            const unsigned long long vmAny = __builtin_amdgcn_ballot_w64(dead != 0xFFFFFFFFu);
                            if (vmAny != 0ull) {
                                const int fl = (int)__builtin_ctzll(vmAny), ll = 63 - (int)__builtin_clzll(vmAny);
                                const unsigned dF = (unsigned)__builtin_amdgcn_readlane((int)dead, fl), dL = (unsigned)__builtin_amdgcn_readlane((int)dead, ll);
                                const int first = 2 * fl + (((dF & 0xFFFFu) == 0u) ? 0 : 1), last = 2 * ll + (((dL >> 16) == 0u) ? 1 : 0);
                                sLowC = min(sLowC, b128 + first); sHighC = max(sHighC, b128 + last);
                                if constexpr (PH == 2) {       // pointer words of the sub-block's first / last unpruned row, for the pre-test of the convergence test
                                    const unsigned cf = (unsigned)__builtin_amdgcn_readlane((int)CS1[r], first >> 1), cl = (unsigned)__builtin_amdgcn_readlane((int)CS1[r], last >> 1);
                                    const int vf = (first & 1) ? (int)cf >> 16 : (int)(short)cf, vl = (last & 1) ? (int)cl >> 16 : (int)(short)cl;
                                    const unsigned aEd = __builtin_amdgcn_inverse_ballot_w64(1ull) ? vcur + O_EDGE + 8u * (unsigned)(R2 * w + r) : vTrashRed;
                                    lds_st<nuc_i2>(aEd, nuc_i2{vf, vl});
                                }
                            }
                            if constexpr (TB) {                                                        // :548-557
                                const unsigned st2 = bfi(gI, 0x00010001u, 0x00020002u) & notM;
                                const unsigned nib = st2 | (IptrM & 0x00040004u) | (DptrM & 0x00080008u);
                                const unsigned sh = 4u * (unsigned)(k & 7);
                                tbA[r] |= (nib & 0xFu) << sh;
                                tbB[r] |= (nib >> 16) << sh;
                            }
                        }
                        LS2[r] = LS1;
                        if constexpr (CONV) LCS2[r] = LCS1;
                        belowS = nbS; belowI = nbI; belowCS = nbCS; belowCI = nbCI; belowOut = nextBelowOut;
                    }
                    fresh = 0;
                    // mailbox: lane 63 of the last sub-block (its odd row is row i - 1 of the next wave's first row)
                    lds_st<nuc_i4>(vcur + mbRel, nuc_i4{(int)S1[R2 - 1], (int)I1[R2 - 1], (int)CS1[R2 - 1], (int)CI1[R2 - 1]});
                } else {
#pragma unroll
                    for (int r = 0; r < R2; ++r) { pOut[r] = 0xFFFFFFFFu; pLeftOut[r] = 0xFFFFFFFFu; }
                }
                if (sHighC >= 0) {      // the band's ends among this wave's rows: one lane posts
                    const unsigned vPost = __builtin_amdgcn_inverse_ballot_w64(1ull) ? vcur : vTrashRed;
                    lds_max_u32_off<O_RED + 4>(vPost, kk16 + (0xFFFFu - (unsigned)sLowC));
                    lds_max_u32_off<O_RED + 8>(vPost, kk16 + (unsigned)sHighC);
                }
                if (__builtin_expect(b + SBR - 1 < Lk, 0)) {    // the super block fell out of the band: take the next one
                    while (SBR * sblk + SBR - 1 < Lk) sblk += W;
                    sel_addr(k);
                    load_q();
                    fresh = 1; tbMust = 1;
                }
                raB += 4u;
                if (raB == lds_off(s_sel) + CAP * 4u) raB = lds_off(s_sel);
                if constexpr (TB) tbPending = true;
                const bool hook = ((k & 7) == 7 || (PH == 1 && k == marker));
                if (hook) {
                    if constexpr (TB) {
                        const int bb = SBR * sblk;
                        const bool live = tbMust != 0 || (bb + SBR - 1 >= tbL0 && bb <= tbU0 + 8);
#pragma unroll
                        for (int r = 0; r < R2; ++r) {
                            if (live) {
                                uint2 *dst = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(512 * (R2 * w + r)) + (unsigned)lane * 8u);
                                *dst = uint2{tbA[r], tbB[r]};
                            }
                            tbA[r] = tbB[r] = 0u;
                        }
                        tbMust = 0; tbL0 = Lk; tbU0 = Uk;
                        tbOff += (unsigned)WINDOW * 4u;
                        tbPending = false;
                    }
                    const int need_hi = ((k + 9 - Lk) >> 6) + 1;
                    if (hiBlk < need_hi) { ++hiBlk; if (w == hiBlk % W) load_ring_block(hiBlk); }
                }
                TWL_SETPRIO(0);
                wg_barrier_lds();

                // ---- post: the band of the next diagonal, :563-604 ----
                const nuc_i4 rd = lds_ld<nuc_i4>(vcur + O_RED);
                msp = max(msp, __builtin_amdgcn_readfirstlane(rd.x));
                const int newL = __builtin_amdgcn_readfirstlane((int)((kk16 + 0xFFFFu) - (unsigned)rd.y));
                const int newU = __builtin_amdgcn_readfirstlane((int)((unsigned)rd.z - kk16));

                if constexpr (CONV) {                                                              // :585-595
                    if (!converged && k < kEnd - 1) {
                        int conv_S = I_UNSET;
                        bool all3 = false;
                        if constexpr (PH == 1) {
                            if (k == marker - 1) conv_S = (newL == newU) ? (int)(short)(0xC000 | (newL & 0x3FFF)) : I_UNSET;
                            else conv_S = (newL == newU) ? (newL & 0x3FFF) : I_UNSET;
                        } else {
                            if (threadIdx.x == 0) {
                                int c0 = 0x7fffffff, c1 = (int)0x80000000, c2 = 0;
                                asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
                                lds_st<nuc_i4>(vprev + O_CONV, nuc_i4{c0, c1, c2, c2});
                            }
                            // pre-test (necessary condition): the two end cells of the surviving band hold the same convergence pointer
                            bool maybe = false;
                            if (newL <= newU) {
                                const int cLo = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newL >> 7) % (unsigned)NSB));
                                const int cHi = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newU >> 7) % (unsigned)NSB) + 4u);
                                maybe = __builtin_amdgcn_readfirstlane(cLo) == __builtin_amdgcn_readfirstlane(cHi);
                            }
                            if (maybe) {
#pragma unroll
                                for (int r = 0; r < R2; ++r) {
                                    const unsigned inR = ~out_mask(r, newL, newU, 0);
                                    const unsigned long long rmE = __builtin_amdgcn_ballot_w64((inR & 0xFFFFu) != 0u), rmO = __builtin_amdgcn_ballot_w64((int)inR < 0);
                                    if ((rmE | rmO) != 0ull) {
                                        const int fE = rmE ? 2 * (int)__builtin_ctzll(rmE) : 0x7fffffff, fO = rmO ? 2 * (int)__builtin_ctzll(rmO) + 1 : 0x7fffffff;
                                        const int first = min(fE, fO);
                                        const unsigned cf = (unsigned)__builtin_amdgcn_readlane((int)CS1[r], first >> 1);
                                        const int v = (first & 1) ? (int)cf >> 16 : (int)(short)cf;
                                        const unsigned V2 = both(v);
                                        const bool badS = __builtin_amdgcn_ballot_w64(((CS1[r] ^ V2) & inR) != 0u) != 0ull;
                                        const bool badID = __builtin_amdgcn_ballot_w64((((CI1[r] ^ V2) | (CD1[r] ^ V2)) & inR) != 0u) != 0ull;
                                        if (lane == 0) {
                                            ds_min_i32_off<O_CONV>(vcur, v);
                                            ds_max_i32_off<O_CONV + 4>(vcur, v);
                                            if (badS || badID) ds_or_b32_off<O_CONV + 8>(vcur, (badS ? 1 : 0) | (badID ? 2 : 0));
                                        }
                                    }
                                }
                                wg_barrier_lds();
                                const nuc_i4 cv = lds_ld<nuc_i4>(vcur + O_CONV);
                                const int vmin = __builtin_amdgcn_readfirstlane(cv.x);
                                const int vmax = __builtin_amdgcn_readfirstlane(cv.y);
                                const int fl = __builtin_amdgcn_readfirstlane(cv.z);
                                if (vmin == vmax && !(fl & 1)) { conv_S = vmin; all3 = !(fl & 2); }
                            }
                        }
                        if (all3 && prev_conv_s == conv_S && conv_S != I_UNSET) { converged = true; conv_value = prev_conv_s; convS = msp; }
                        prev_conv_s = conv_S;
                    }
                }
                {                                                                                  // :597-604
                    sLo2p = sLo1 + 1; sW2 = sW1; sLo1 = Lk; sW1 = width1; vwid1 = width1 + 1;
                    Lk = max(max(newL, k + 2 - refLen), 0);
                    Uk = min(newU + 1, qLen - 1);
                    vcur ^= parx; vprev ^= parx;
                }
                bool ended = false;
                if constexpr (CONV) {
                    if (converged && max(msp, 0) > convS) { conv_logic = true; go_on = false; ended = true; }      // :607-612
                }
                if (!ended) {
                    ++k;
                    if (__builtin_expect((unsigned)(Uk - Lk) >= (unsigned)fcap, 0)) {
                        if (k >= kEnd) {}
                        else if (Lk > Uk) { tile_err = 1; go_on = false; }
                        else if (Uk - Lk + 1 > fLen) { tile_err = 2; go_on = false; }
                        else if (Uk / SBR - Lk / SBR >= W) { tile_err = kErrOverflow; go_on = false; }      // it really outgrew this window
                        else if (SBR * sblk + SBR - 1 < Lk) {
                            while (SBR * sblk + SBR - 1 < Lk) sblk += W;
                            sel_addr(k); load_q(); fresh = 1; tbMust = 1;
                        }
                    }
                }
            };

            if (steps_left < 0) { tile_err = 3; go_on = false; }
            {
                using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
                const int kA = min(kEnd, marker - 1);
                while (go_on && k < kA) step(T0{});
                const int kB = min(kEnd, marker + 1);
#pragma unroll
                for (int r = 0; r < R2; ++r) { CS1[r] = both(I_UNSET); CI1[r] = both(I_IB); CD1[r] = both(I_DB); LCS2[r] = both(I_UNSET); }
                while (go_on && k < kB) step(T1{});
                const int kCap = min(kEnd, kI16MaxDiag);      // scores must stay 16-bit
                while (go_on && k < kCap) step(T2{});
                if (go_on && k < kEnd) { tile_err = kErrGuard; go_on = false; }
            }
            const int last_k = conv_logic ? k : k - 1;
            steps_left -= (long long)(last_k + 1);
            cells += vcells;
            dbg_lastk = last_k; dbg_conv = conv_value; dbg_L = Lk; dbg_U = Uk;
            if (tile_err != 0) { err = tile_err; break; }
            {      // a letter outside A, C, G, T (or a profile that is not one sequence): the fp32 kernel re-runs the pair
                if (guardBad) s_misc[4] = 1;
                __syncthreads();
                guardBad = __builtin_amdgcn_readfirstlane(s_misc[4]) != 0;
                if (guardBad) { err = kErrGuard; break; }
            }
            if (tbPending) {
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    uint2 *dst = reinterpret_cast<uint2 *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(512 * (R2 * w + r)) + (unsigned)lane * 8u);
                    *dst = uint2{tbA[r], tbB[r]};
                }
            }

            // ---- tile exit, :615-682 ----
            int conv_q = 0, conv_r = 0, tb_state = 0, start_k = 0;
            bool bad = false;
            if (!conv_logic && last_k >= marker) {                            // :633-635 needs CS[last_k][0]: the pointer word of row sLo1 (the band of diagonal last_k)
                const int Llast = sLo1;
#pragma unroll
                for (int r = 0; r < R2; ++r) {
                    const int b128 = SBR * sblk + 128 * r;
                    if (Llast >= b128 && Llast <= b128 + 127 && lane == ((Llast - b128) >> 1)) s_misc[1] = ((Llast - b128) & 1) ? (int)CS1[r] >> 16 : (int)(short)CS1[r];
                }
                __syncthreads();
                conv_value = __builtin_amdgcn_readfirstlane(s_misc[1]);
            }
            if (conv_logic || last_k >= marker) {
                if (conv_value == I_UNSET || conv_value == I_IB || conv_value == I_DB) bad = true;      // boundary sentinel / unset: the reference indexes out of range here
                else {
                    conv_q = conv_value & 0x3FFF;
                    tb_state = (conv_value >> 14) & 3;
                    conv_r = marker - conv_q - ((tb_state == 3) ? 1 : 0);
                    start_k = (tb_state == 3) ? marker - 1 : marker;
                    if (conv_r < 0) bad = true;
                }
            } else {                                                          // :625-632
                conv_q = qLen - 1; conv_r = refLen - 1; start_k = last_k; tb_state = 0; last_tile = true;
            }
            if (bad) { err = 3; break; }
            ref_idx += conv_r; qry_idx += conv_q;                             // :654-655
            if (R - ref_idx < 0 || Q - qry_idx < 0) { err = 3; break; }       // :659-668
            int tailDir = 0, tailLen = 0;
            if (ref_idx == R - 1 && qry_idx < Q - 1) { tailDir = 1; tailLen = Q - qry_idx - 1; last_tile = true; }   // :671-674
            if (qry_idx == Q - 1 && ref_idx < R - 1) { tailDir = 2; tailLen = R - ref_idx - 1; last_tile = true; }   // :675-678
            if (ref_idx == R - 1 && qry_idx == Q - 1) last_tile = true;       // :679

            __syncthreads();   // all traceback-pointer stores of this tile are complete and visible
            if (w == 0) {
                int n = 0;
                {   // Traceback, :134-231 (as in talco_lean_kernel: LDS patches of 64 rows x 16 groups of 8 diagonals; the selector ring and the CD mirrors are dead until the next tile)
                    constexpr int PG = 16;
                    static_assert(sizeof(s_par) >= PG * 64 * sizeof(uint32_t), "traceback patch lives in the parity buffers");
                    uint32_t *s_patch = reinterpret_cast<uint32_t *>(&s_par[0]);
                    int kk2 = start_k, ii = conv_q, qi = conv_q, ri = conv_r, st = tb_state % 3;
                    const bool first = (tile == 0);
                    bool done = (kk2 < 0);
                    while (!done) {
                        const int g0 = kk2 >> 3, i0 = ii;
                        const int row = i0 - 63 + lane;
#pragma unroll
                        for (int t = 0; t < PG; ++t) {
                            uint32_t word = 0u;
                            if (g0 - t >= 0 && row >= 0)
                                word = __hip_atomic_load(&tb[(size_t)(g0 - t) * WINDOW + (size_t)((unsigned)row % (unsigned)WINDOW)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_patch[t * 64 + lane] = word;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) {
                            for (;;) {
                                const int t = g0 - (kk2 >> 3), l = 63 - (i0 - ii);
                                if (t >= PG || l < 0) break;
                                const uint32_t word = s_patch[t * 64 + l];
                                const int v = (int)((word >> (4 * (kk2 & 7))) & 0xFu);
                                int dir;
                                if (st == 0) {
                                    st = v & 3;
                                    if (st == 0) dir = 0;
                                    else if (st == 1) { dir = 1; st = (v & 4) ? 1 : 0; }
                                    else { dir = 2; st = (v & 8) ? 2 : 0; }
                                } else if (st == 1) { dir = 1; st = (v & 4) ? 1 : 0; }
                                else { dir = 2; st = (v & 8) ? 2 : 0; }
                                if (dir == 0) { kk2 -= 2; ii -= 1; qi--; ri--; }
                                else if (dir == 1) { kk2 -= 1; ii -= 1; qi--; }
                                else { kk2 -= 1; ri--; }
                                s_rev[n++] = (int8_t)dir;
                                if (kk2 < 0) { done = true; break; }
                                if (first && (ri < 0 || qi < 0)) { done = true; break; }
                                if (ii < 0) { done = true; break; }
                            }
                        }
                        kk2 = __builtin_amdgcn_readfirstlane(kk2);
                        ii = __builtin_amdgcn_readfirstlane(ii);
                        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (lane == 0 && first) {
                        while (ri > -1) { s_rev[n++] = 2; ri--; }
                        while (qi > -1) { s_rev[n++] = 1; qi--; }
                    }
                }
                n = __builtin_amdgcn_readfirstlane(n);
                const int skip = (tile > 0) ? 1 : 0;                          // :98-102
                const int cnt = n - skip;
                if (pos + cnt + tailLen > 2 * a.seq_len) { err = 3; }
                else {
                    for (int t = lane; t < cnt; t += 64) out[pos + t] = s_rev[n - 1 - skip - t];
                    for (int t = lane; t < tailLen; t += 64) out[pos + cnt + t] = (int8_t)tailDir;
                    pos += cnt + tailLen;
                }
                if (lane == 0) { s_misc[2] = err; s_misc[3] = pos; }
            }
            __syncthreads();
            err = __builtin_amdgcn_readfirstlane(s_misc[2] == 3 ? 3 : err);
            pos = __builtin_amdgcn_readfirstlane(s_misc[3]);
            if (err != 0) break;
            tile += 1;
        }

        __syncthreads();
        if (threadIdx.x == 0) {
            a.err[pair] = (int16_t)err;
            a.aln_len[pair] = (err == 0) ? pos : 0;
            a.cells[pair] = cells;
            if (a.dbg) {
                int32_t *g = a.dbg + 16 * (size_t)pair;
                g[0] = tile; g[1] = dbg_lastk; g[2] = dbg_conv; g[3] = dbg_L; g[4] = dbg_U; g[5] = ref_idx; g[6] = qry_idx;
                g[7] = pos; g[8] = err; g[9] = (int)min(steps_left, 0x7fffffffll); g[10] = R; g[11] = Q;
            }
        }
        __syncthreads();
    }
}

}  // namespace twl

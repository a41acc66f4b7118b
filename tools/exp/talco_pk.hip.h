// twilight_amd/csrc/talco_pk.hip.h -- the nucleotide TALCO-XDrop kernel of round 5: TWO query rows per lane, the column score and the gap sums as PACKED fp32.
//
// What it computes: Talco_xdrop::Align_freq / Tile / Traceback (/root/reference/src/TALCO-XDrop.cpp:62-108, :233-689, :134-231), fp32, x86 TALCO_SIMD
// operation order -- bit for bit what talco_lean_kernel (talco_nuc.hip.h) computes; the parity suite holds the two to each other and to the oracle.
//
// Why.  A wave issues one instruction of any kind per ~4.5 cycles, and of the ~170 instructions of a 64-row block step ~75 are the fp32 multiplies and adds
// of the column score (:378-395), the gap-letter terms (:394-395), the division (:444) and the gap sums (:456-463).  gfx950 has v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two fp32 operations per issue slot, each rounded exactly like the plain form.  Packing WITHIN a cell was tried in round 4 and lost (register
// pairs have to be shuffled together).  Packing ACROSS two cells needs no shuffle when the two cells are the same lane's: a lane owns rows i and i + 64 of a
// 128-row block, every per-row quantity is a register pair {row i, row i + 64}, and one packed instruction advances both.  What does not pack (compares,
// selects, pointer bits) is done per half as before; what is per block (mailbox, ring address, activity test, priorities, posts of the band's ends) is done
// once per 128 rows instead of once per 64.  ~190 instructions per 128 cells against ~340.
//
// What had to change for it:
//   * the reference ring is LETTER-MAJOR: plane t holds float t of every staged column (A, C, G, T, N, gap, gapOpen, gapExtend), so that the two columns a lane
//     needs on a diagonal -- j = k - i and j - 64 -- are 64 floats apart in every plane and ONE ds_read2st64_b32 returns them as an aligned register pair
//     (columns in decreasing order, so that the pair is {lower half, upper half}); slots 0..63 are kept once more behind the end so that a pair never straddles the wrap;
//   * row i - 1 of the upper half's lane 0 is the lower half's lane 63: a wave_ror:1 of the lower half supplies it as the `old` operand of the upper half's
//     wave_shr:1 (three DPP moves for the two halves of a value instead of two);
//   * the band tests of the two older diagonals are the scalar lane masks the previous diagonal made (a row is in band(k-1) iff it was in band(k) one step
//     ago), moved up a lane by scalar shifts for the "row i - 1" forms; only "row in band(k)" is compared afresh;
//   * the first and last unpruned row of the wave's rows are found on the scalar unit (s_ff1 / s_flbit of the survivors' mask) and posted once per wave.
// Geometry: W waves, one 128-row block each; window 128 W rows.  Bands are guaranteed to fit up to 128 (W - 1) + 1 rows.
#pragma once
#include "talco_nuc.hip.h"

namespace twl {

__device__ __forceinline__ unsigned long long lane_mask(int lo, int hi)
{
    const int a = max(lo, 0), b = min(hi, 63);
    const unsigned long long m = (~0ull << a) & (~0ull >> (63 - b));
    return (a <= b) ? m : 0ull;
}
template <int W>
struct PkCfg {
    static constexpr int NV = 2 * W;             // 64-row half blocks resident at once
    static constexpr int WINDOW = 64 * NV;       // rows
    static constexpr int NB = NV + 2;            // 64-column ring stages
    static constexpr int CAP = 64 * NB;          // ring columns
    static constexpr int PITCH = CAP + 64;       // floats per letter plane: columns 0..63 once more behind the end
    static constexpr int UNIT = PITCH / 64;      // plane pitch in units of 64 floats (the offset unit of ds_read2st64_b32)
    static constexpr int THREADS = 64 * W;
};

// {mem[addr + 256 * U0], mem[addr + 256 * U1]}: two floats 64-float units apart as one register pair (the compiler makes it ONE ds_read2st64_b32)
template <int U0, int U1>
__device__ __forceinline__ nuc_f2 lds_pair64(unsigned addr)
{
    const __attribute__((address_space(3))) float *p = (const __attribute__((address_space(3))) float *)((const __attribute__((address_space(3))) char *)nullptr + addr);
    return nuc_f2{p[64 * U0], p[64 * U1]};
}

__device__ __forceinline__ float dpp_ror1_f(float src)      // lane t <- src[t-1], lane 0 <- src[63] (DPP wave_ror:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x13C, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_ror1_i(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x13C, 0xf, 0xf, false); }
__device__ __forceinline__ nuc_f2 pk_fma(nuc_f2 a, nuc_f2 b, nuc_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// MM = 2 (match / transition / transversion structure of the matrix, zero N row and column) or 5 (mode 2 for query rows with one non-zero letter); MT as in
// talco_lean_kernel: 0 pairs from a queue, 1 one tile per job from its predicted start, 2 scout of a tile boundary.
template <int MM, int W, int MINW, int MT>
__global__ __launch_bounds__(64 * W, MINW) void talco_pk_kernel(NArgs a)
{
    static_assert(MM == 2 || MM == 5, "column-score mode");
    static_assert(MT >= 0 && MT <= 2, "pairs, tile jobs or scouts");
    constexpr int P = 6;
    constexpr bool GUESS = (MT == 2);
    using C = PkCfg<W>;
    constexpr int NV = C::NV, WINDOW = C::WINDOW, NB = C::NB, CAP = C::CAP, PITCH = C::PITCH, UNIT = C::UNIT;
    constexpr int F4 = 2;

    __shared__ float s_ring[8 * PITCH];
    struct ParBuf {
        int cd[WINDOW + 4];          // offset-addressed mirror of the reference's CD rows (see talco_kernel)
        int4 exch[W];                // mailbox: lane 63 of a block's upper half -> lane 0 of the next block's lower half {S, I, CS, CI}
        unsigned red[4];             // {running max S (float bits), low-end tag, high-end tag, -}
        int conv[4];                 // {vmin, vmax, flags, -}
        int edge[2 * NV];            // phase C: CS of the first / last unpruned row of every 64-row half (the cheap pre-test of the convergence test)
        int4 trash[64];              // per-lane trash slots: single-lane LDS side effects without touching EXEC
    };
    __shared__ ParBuf s_par[2];
    __shared__ int s_misc[8];
    __shared__ int8_t s_rev[2 * kMaxMarker + 16];
    __shared__ unsigned long long s_team[4];      // scouts: {-, -, best cell of diagonal marker-1, of diagonal marker}
    constexpr unsigned O_CD = (unsigned)offsetof(ParBuf, cd), O_EXCH = (unsigned)offsetof(ParBuf, exch), O_RED = (unsigned)offsetof(ParBuf, red),
                       O_CONV = (unsigned)offsetof(ParBuf, conv), O_TRASH = (unsigned)offsetof(ParBuf, trash), O_EDGE = (unsigned)offsetof(ParBuf, edge);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tb = a.tb + (size_t)blockIdx.x * (size_t)a.tb_words;
    const float inf = (float)(2.0 * (double)a.xdrop + 1.0);     // TALCO-XDrop.cpp:252
    const float xdropf = (float)a.xdrop;

    for (;;) {
        if (threadIdx.x == 0) s_misc[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_misc[0]);
        if (item >= a.n_items) break;
        const int pair = __builtin_amdgcn_readfirstlane((MT == 1 || MT == 2) ? a.mt_jobs[3 * item] : a.items[item]);
        const int slot = (MT == 1 || MT == 2) ? __builtin_amdgcn_readfirstlane(a.mt_jobs[3 * item + 1]) : 0;
        const int mtx = (MT == 1 || MT == 2) ? __builtin_amdgcn_readfirstlane(a.mt_jobs[3 * item + 2]) : item;
        const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
        const float gc = (a.gc_zero && __builtin_amdgcn_readfirstlane((int)a.gc_zero[pair])) ? 0.0f : a.gap_char;
        const bool gcNZ = (gc != 0.0f);
        const int spLo = (a.marker - 1) * slot - 1, spHi = a.marker * slot + 1;
        const int scoutD0 = max(spLo - a.mt_lead, 0);
        const int marker = (MT == 2) ? min(spHi + a.mt_marg - scoutD0, kMaxMarker) : a.marker;
        const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];   // :255,:269
        const bool denomOne = (denom == 1.0f);
        const float rden = refined_rcp(denom);
        const float4 *colsR = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * (P + 2));
        const float4 *colsQ = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * (P + 2));

        int ref_idx = 0, qry_idx = 0, tile = 0, pos = 0, err = 0;
        bool last_tile = (R <= 0 || Q <= 0);
        if constexpr (MT == 1) {          // one tile, from its predicted start (tile 0: from the corner)
            if (slot > 0) {
                ref_idx = __builtin_amdgcn_readfirstlane(a.mt_chain[((size_t)mtx * a.mt_slots + slot) * 2]);
                qry_idx = __builtin_amdgcn_readfirstlane(a.mt_chain[((size_t)mtx * a.mt_slots + slot) * 2 + 1]);
                tile = 1;
                if (ref_idx < 0) last_tile = true;      // no prediction for this tile
            }
            const int32_t *fr = a.mt_front + (size_t)mtx * 8;
            const int fs = __builtin_amdgcn_readfirstlane(fr[0]), ft = __builtin_amdgcn_readfirstlane(fr[1]);
            const int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + slot) * kMtRec;
            const int rv = __builtin_amdgcn_readfirstlane(rc[0]), rr = __builtin_amdgcn_readfirstlane(rc[1]), rq = __builtin_amdgcn_readfirstlane(rc[2]);
            if (fs == 2 || (fs == 1 && slot < ft) || (rv == 1 && rr == ref_idx && rq == qry_idx)) last_tile = true;
        }
        if constexpr (MT == 2) {          // the cell of diagonal scoutD0 on the straight line between the corners
            tile = 1;
            if (spLo > R + Q - 2 || R < 2 || Q < 2) last_tile = true;
            else {
                qry_idx = (int)(((long long)scoutD0 * Q) / (R + Q));
                qry_idx = min(qry_idx, Q - 1);
                ref_idx = scoutD0 - qry_idx;
                if (ref_idx > R - 1) { ref_idx = R - 1; qry_idx = scoutD0 - ref_idx; }
                if (qry_idx > Q - 1 || ref_idx < 0) last_tile = true;
            }
        }
        const int jobRef = ref_idx, jobQry = qry_idx;
        int8_t *out = (MT == 1) ? a.mt_seg + ((size_t)mtx * a.mt_slots + slot) * (size_t)a.mt_segcap : a.aln + (size_t)pair * 2 * (size_t)a.seq_len;
        unsigned long long cells = 0;
        long long steps_left = (long long)(R + Q + 2) * ((R + Q) / (max(marker, 2) - 1) + 4) + a.step_slack;
        if (!last_tile && !(denom >= 1.0f && denom <= 1.0995116e12f)) { err = kErrGuard; last_tile = true; }
        int dbg_lastk = 0, dbg_conv = 0, dbg_L = 0, dbg_U = 0;
        bool guardBad = false;

        while (!last_tile) {   // ---- Align_freq tile loop, TALCO-XDrop.cpp:77-106 ----
            const int refLen = R - ref_idx, qLen = Q - qry_idx;
            const int fLen = min(a.flen, min(refLen, qLen));                          // :258
            const int fcap = min(fLen, 64 * (NV - 2));
            // ---- the wave's block: lower half = rows 128 dblk + lane (.x of every pair), upper half = 64 rows above it (.y) ----
            nuc_f2 S1, I1, D1, LS2;
            int CS1[2], CI1[2], CD1[2], LCS2[2];
            nuc_f2 q2[P], gopq2, gexq2;
            nuc_f2 qT2[4];                 // MM 5: the row sums of the first products (see talco_lean_kernel)
            int dblk;
            unsigned raA;                  // byte address, plane 0, of the LOWER half's reference column j = k - i (the upper half's, j - 64, sits 64 floats further: the ring
                                           // holds the columns in DEcreasing order, so that a pair comes out of ds_read2st64_b32 as {lower half, upper half})
            uint32_t tbacc[2];
            bool q5any;
            // scalar lane masks of the two halves on the previous diagonal: rows in band(k-1); rows whose upper neighbour i-1 is in band(k-1)
            unsigned long long pIn[2], pLeft[2];
            int fresh;      // (an int: as a bool captured by the step lambda it stayed in memory and took the loop's uniformity with it)

            auto load_q = [&]() __attribute__((always_inline)) {
                float cb[2][8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 128 * dblk + 64 * h + lane;
                    const bool ok = qry_idx + i < Q;
#pragma unroll
                    for (int t = 0; t < F4; ++t) {
                        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) c = colsQ[F4 * (size_t)(qry_idx + i) + t];
                        cb[h][4 * t] = c.x; cb[h][4 * t + 1] = c.y; cb[h][4 * t + 2] = c.z; cb[h][4 * t + 3] = c.w;
                    }
                }
#pragma unroll
                for (int t = 0; t < P; ++t) q2[t] = nuc_f2{cb[0][t], cb[1][t]};
                gopq2 = nuc_f2{cb[0][P], cb[1][P]}; gexq2 = nuc_f2{cb[0][P + 1], cb[1][P + 1]};
                if constexpr (MM == 5) {      // one non-zero letter per query row: qT[l] = the row sum of the first products (= q[m*] * M[l][m*]), :386
#pragma unroll
                    for (int l = 0; l < 4; ++l) qT2[l] = ((q2[0] * a.M[5 * l + 0] + q2[1] * a.M[5 * l + 1]) + q2[2] * a.M[5 * l + 2]) + q2[3] * a.M[5 * l + 3];
                }
                q5any = gcNZ && __builtin_amdgcn_ballot_w64(cb[0][P - 1] != 0.0f || cb[1][P - 1] != 0.0f) != 0ull;
                bool bad = false;
#pragma unroll
                for (int t = 0; t < P; ++t) bad = bad | div_guard_bad(cb[0][t]) | div_guard_bad(cb[1][t]);
                guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
            };
            auto load_ring_block = [&](int B) __attribute__((always_inline)) {      // 64 reference columns into their letter planes
                const int col = 64 * B + lane;
                const int st = B % NB;
                const int sl = CAP - 1 - (st * 64 + lane);      // column c lives in slot CAP - 1 - (c mod CAP)
                float f[8];
#pragma unroll
                for (int t = 0; t < F4; ++t) {
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (col < refLen) c = colsR[F4 * (size_t)(ref_idx + col) + t];
                    f[4 * t] = c.x; f[4 * t + 1] = c.y; f[4 * t + 2] = c.z; f[4 * t + 3] = c.w;
                }
                bool bad = false;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    s_ring[t * PITCH + sl] = f[t];
                    if (st == NB - 1) s_ring[t * PITCH + CAP + sl] = f[t];      // (slots 0..63 once more behind the end)
                    if (t < P) bad = bad | div_guard_bad(f[t]);
                }
                guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
            };
            auto ring_addr = [&](int k) __attribute__((always_inline)) {      // slot of column j = k - 128 dblk - lane: CAP - 1 - (j mod CAP)
                int rs = (k - 128 * dblk - lane) % CAP;
                rs += (rs < 0) ? CAP : 0;
                raA = (unsigned)(CAP - 1 - rs) * 4u + lds_off(s_ring);
            };

            dblk = w;
            ring_addr(0);
            tbacc[0] = tbacc[1] = 0;
            S1 = I1 = D1 = LS2 = nuc_f2{-1.0f, -1.0f};
            load_q();
            pIn[0] = pIn[1] = pLeft[0] = pLeft[1] = 0ull; fresh = 0;      // (bands k-1 and k-2 are empty when the tile begins)
            int hiBlk = 1;
            if (w == 0 % W) load_ring_block(0);
            if (w == 1 % W) load_ring_block(1);
            for (int t = threadIdx.x; t < WINDOW + 4; t += C::THREADS) { s_par[0].cd[t] = kDB; s_par[1].cd[t] = kDB; }      // :308
            if (threadIdx.x == 0) {
                s_par[0].red[0] = s_par[1].red[0] = (unsigned)__float_as_int(-inf);
                s_par[0].red[1] = s_par[0].red[2] = s_par[1].red[1] = s_par[1].red[2] = 0u;
                for (int t = 0; t < 2; ++t) { s_par[t].conv[0] = 0x7fffffff; s_par[t].conv[1] = (int)0x80000000; s_par[t].conv[2] = 0; }
                s_team[2] = 0ull; s_team[3] = 0ull;
                s_misc[4] = 0;
            }
            __syncthreads();

            // ---- Tile, TALCO-XDrop.cpp:233-689 (band bookkeeping as in talco_lean_kernel: uniform vectors + the scalars Lk / Uk, and the scalar twins of the older bands) ----
            int vL = 0, vU = 0, vlo1 = 0x3fffffff;
            unsigned vw1 = 0;
            int vwid1 = 0;
            unsigned vcells = 0;
            int Lk = 0, Uk = 0;
            int sLo1 = 0x3fffffff, sLo2p = 0x3fffffff;
            unsigned sW1 = 0, sW2 = 0;
            float msp = -inf, convf = 0.0f;
            bool converged = false;
            int conv_value = 0, prev_conv_s = -1;
            const int kEnd = refLen + qLen - 1;
            int k = 0;
            int tile_err = 0;
            bool go = true, conv_logic = false;
            bool spec = true;
            bool tbPending = false;
            unsigned tbOff = (unsigned)lane * 4u;
            unsigned vcur = lds_off(&s_par[0]), vprev = lds_off(&s_par[1]);
            const unsigned parx = lds_off(&s_par[0]) ^ lds_off(&s_par[1]);
            const unsigned relTrashRed = O_TRASH + (unsigned)lane * 4u - O_RED;
            const unsigned mbRel = (lane == 63) ? O_EXCH + 16u * (unsigned)w : O_TRASH + (unsigned)lane * 16u;
            const unsigned exRel = O_EXCH + 16u * (unsigned)((w + W - 1) % W);

            auto step = [&](auto PHtag) __attribute__((always_inline)) {
                constexpr int PH = decltype(PHtag)::value;
                constexpr bool TB = (PH != 2), CONV = (PH != 0);
                const unsigned kk16 = (unsigned)(k + 1) << 16;
                const int width1 = Uk - Lk;
                const unsigned vwidth1 = (unsigned)(vU - vL);
                vcells += vwidth1 + 1u;
                const float thr = __int_as_float(max(__float_as_int(msp), 0)) - xdropf;   // :495 with :607
                bool special = false;
                if (__builtin_expect(spec, 0)) {
                    special = (k == 0) | ((tile == 0) && (Lk == 0 || Uk == k));
                    spec = special;
                }
                int staleCD = kDB;
                if constexpr (PH == 2) staleCD = lds_ld<int>(vprev + 4u * (unsigned)vwid1 + O_CD);
                const unsigned vTrashRed = vcur + relTrashRed;
                const int b = 128 * dblk;
                // the block takes part when the band touches it or will reach its first row on the next diagonal: Lk - 127 <= b <= Uk + 1
                if ((unsigned)(b - (Lk - 127)) <= (unsigned)(width1 + 128)) {
                    TWL_SETPRIO(2);
                    // ---- loads: mailbox of the previous block, reference columns of the two cells ----
                    float eS, eI; int eCS = 0, eCI = 0;
                    if constexpr (CONV) {
                        const nuc_i4 e = lds_ld<nuc_i4>(vprev + exRel);
                        eS = __int_as_float(e.x); eI = __int_as_float(e.y); eCS = e.z; eCI = e.w;
                    } else {
                        const nuc_i2 e = lds_ld<nuc_i2>(vprev + exRel);
                        eS = __int_as_float(e.x); eI = __int_as_float(e.y);
                    }
                    // {lower half's column j = k - i, upper half's column j - 64} of every letter plane
                    const nuc_f2 r0 = lds_pair64<0 * UNIT, 0 * UNIT + 1>(raA), r1 = lds_pair64<1 * UNIT, 1 * UNIT + 1>(raA);
                    const nuc_f2 r2 = lds_pair64<2 * UNIT, 2 * UNIT + 1>(raA), r3 = lds_pair64<3 * UNIT, 3 * UNIT + 1>(raA);
                    const nuc_f2 gopr = lds_pair64<6 * UNIT, 6 * UNIT + 1>(raA), gexr = lds_pair64<7 * UNIT, 7 * UNIT + 1>(raA);
                    nuc_f2 rN = nuc_f2{0.f, 0.f}, rg = nuc_f2{0.f, 0.f};
                    if (gcNZ) { rN = lds_pair64<4 * UNIT, 4 * UNIT + 1>(raA); rg = lds_pair64<5 * UNIT, 5 * UNIT + 1>(raA); }
                    // ---- the band tests as scalar lane masks (h = 0 lower half, 1 upper half) ----
                    unsigned long long mUp[2], mLeft[2], mDiag[2], mIn[2];
                    if (__builtin_expect(fresh != 0, 0)) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int bh = b + 64 * h;
                            mUp[h] = lane_mask(sLo1 - bh, sLo1 - bh + (int)sW1);
                            mLeft[h] = lane_mask(sLo1 + 1 - bh, sLo1 + 1 - bh + (int)sW1);
                            mDiag[h] = lane_mask(sLo2p - bh, sLo2p - bh + (int)sW2);
                        }
                        fresh = 0;
                    } else {
                        // row b - 1 in band(k-1): the sign of a 64-bit difference (plain scalar arithmetic)
                        const long long dlt = (long long)(unsigned long long)sW1 - (long long)(unsigned long long)(unsigned)(b - 1 - sLo1);
                        mUp[0] = pIn[0]; mUp[1] = pIn[1];
                        mLeft[0] = (pIn[0] << 1) | ((unsigned long long)(~dlt) >> 63);
                        mLeft[1] = (pIn[1] << 1) | (pIn[0] >> 63);
                        mDiag[0] = pLeft[0]; mDiag[1] = pLeft[1];
                    }
                    {
                        const unsigned tA = (unsigned)(lane - (Lk - b));
                        mIn[0] = __builtin_amdgcn_ballot_w64(tA <= (unsigned)width1);
                        mIn[1] = __builtin_amdgcn_ballot_w64(tA + 64u <= (unsigned)width1);
                    }
                    pIn[0] = mIn[0]; pIn[1] = mIn[1]; pLeft[0] = mLeft[0]; pLeft[1] = mLeft[1];
                    // ---- column score, :378-395, both cells at once: t[l][m] = (q[m]*M[l][m]) * r[l]; s_l = ((t0 + t1) + t2) + t3; numer = ((s0 + s1) + s2) + s3 ----
                    nuc_f2 numer;
                    if constexpr (MM == 5) {
                        numer = ((qT2[0] * r0 + qT2[1] * r1) + qT2[2] * r2) + qT2[3] * r3;
                    } else {
                        const float mA = a.M[0], mB = a.M[2], mC = a.M[1];
                        auto fp = [&](int m, int l) __attribute__((always_inline)) { return q2[m] * ((l == m) ? mA : (((l ^ m) == 2) ? mB : mC)); };
                        const nuc_f2 s0 = ((fp(0, 0) * r0 + fp(1, 0) * r0) + fp(2, 0) * r0) + fp(3, 0) * r0;
                        const nuc_f2 s1 = ((fp(0, 1) * r1 + fp(1, 1) * r1) + fp(2, 1) * r1) + fp(3, 1) * r1;
                        const nuc_f2 s2 = ((fp(0, 2) * r2 + fp(1, 2) * r2) + fp(2, 2) * r2) + fp(3, 2) * r2;
                        const nuc_f2 s3 = ((fp(0, 3) * r3 + fp(1, 3) * r3) + fp(2, 3) * r3) + fp(3, 3) * r3;
                        numer = ((s0 + s1) + s2) + s3;
                    }
                    // the gap-letter terms, :394-395 (every one of them is +-0 when gapCharScore is 0)
                    if (q5any) {
                        numer += (r0 * q2[5]) * gc; numer += (r1 * q2[5]) * gc; numer += (r2 * q2[5]) * gc; numer += (r3 * q2[5]) * gc; numer += (rN * q2[5]) * gc;      // :394
                    }
                    if (gcNZ && (((mIn[0] & __builtin_amdgcn_ballot_w64(rg.x != 0.0f)) | (mIn[1] & __builtin_amdgcn_ballot_w64(rg.y != 0.0f))) != 0ull)) {
#pragma unroll
                        for (int m = 0; m < 5; ++m) numer += (rg * q2[m]) * gc;             // :395
                    }
                    nuc_f2 sim = numer;                                                        // :444
                    if (!denomOne) {
                        const nuc_f2 d2 = nuc_f2{denom, denom}, rr = nuc_f2{rden, rden};
                        const nuc_f2 q0 = numer * rr;
                        const nuc_f2 t0 = pk_fma(-d2, q0, numer);
                        const nuc_f2 q1 = pk_fma(t0, rr, q0);
                        const nuc_f2 t1 = pk_fma(-d2, q1, numer);
                        sim = pk_fma(t1, rr, q1);
                    }
                    // ---- neighbours: row i-1 of the lower half = the lane below (lane 0: the mailbox); of the upper half = the lane below, lane 0: the lower half's lane 63 ----
                    nuc_f2 LS1, LI1;
                    LS1.x = dpp_shr1_f(eS, S1.x); LS1.y = dpp_shr1_f(dpp_ror1_f(S1.x), S1.y);
                    LI1.x = dpp_shr1_f(eI, I1.x); LI1.y = dpp_shr1_f(dpp_ror1_f(I1.x), I1.y);
                    int LCS1[2] = {0, 0}, LCI1[2] = {0, 0};
                    if constexpr (CONV) {
                        LCS1[0] = dpp_shr1_i(eCS, CS1[0]); LCS1[1] = dpp_shr1_i(dpp_ror1_i(CS1[0]), CS1[1]);
                        LCI1[0] = dpp_shr1_i(eCI, CI1[0]); LCI1[1] = dpp_shr1_i(dpp_ror1_i(CI1[0]), CI1[1]);
                    }
                    // ---- recurrence, :445-497: the sums packed, the choices per half ----
                    const nuc_f2 matchS = LS2 + sim;
                    const nuc_f2 delOp = S1 + gopr, delExt = D1 + gexr;                        // :456-463
                    const nuc_f2 insOp = LS1 + gopq2, insExt = LI1 + gexq2;
                    float Sn[2], In[2], Dn[2];
                    unsigned long long vmAll[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int i = b + 64 * h + lane;
                        const bool up_ok = __builtin_amdgcn_inverse_ballot_w64(mUp[h]);
                        const bool left_ok = __builtin_amdgcn_inverse_ballot_w64(mLeft[h]);
                        const bool diag_ok = __builtin_amdgcn_inverse_ballot_w64(mDiag[h]);
                        const bool inband = __builtin_amdgcn_inverse_ballot_w64(mIn[h]);
                        float match = diag_ok ? matchS[h] : -inf;
                        if (__builtin_expect(special, 0)) {
                            const int j = k - i;
                            if (k == 0) match = sim[h];
                            else if (i == 0 || j == 0) {
                                int far = max(i, j) - 1; far = far < 0 ? 0 : far;
                                match = (sim[h] + a.gap_open) + a.gap_extend * (float)far;
                            }
                        }
                        const bool dGE = delExt[h] >= delOp[h], iGE = insExt[h] >= insOp[h];
                        const bool Dptr = !up_ok | dGE;
                        const bool Iptr = !left_ok | iGE;
                        const float Dv = up_ok ? (dGE ? delExt[h] : delOp[h]) : -inf;
                        const float Iv = left_ok ? (iGE ? insExt[h] : insOp[h]) : -inf;
                        const bool gapIsI = Iv > Dv;                                            // :477-494
                        const float Gv = gapIsI ? Iv : Dv;
                        const bool isM = match >= Gv;
                        float Sv = isM ? match : Gv;
                        // :495-497 and the band in one select (a surviving cell has Sv >= thr > -inf: the survivors ARE the unpruned rows)
                        const unsigned long long vm = __builtin_amdgcn_ballot_w64(Sv >= thr) & mIn[h];
                        Sv = __builtin_amdgcn_inverse_ballot_w64(vm) ? Sv : -inf;
                        vmAll[h] = vm;
                        if constexpr (CONV) {                                                      // :520-547
                            int CSn, CIn, CDn;
                            const int i16 = i & 0xFFFF;
                            if (PH == 1 && k == marker - 1) { CSn = (3 << 16) | i16; CIn = CI1[h]; CDn = CD1[h]; }
                            else if (PH == 1) { CSn = i16; CIn = (1 << 16) | i16; CDn = (2 << 16) | i16; }
                            else {
                                const int viaS = (LCS1[h] != -1) ? LCS1[h] : kIB;
                                CIn = left_ok ? (Iptr ? LCI1[h] : viaS) : kIB;
                                const int cdUp = up_ok ? CD1[h] : staleCD;      // above the stored band the reference reads the stale slot (:535)
                                CDn = Dptr ? cdUp : ((CS1[h] != -1) ? CS1[h] : kDB);
                                const int viaGap = gapIsI ? CIn : CDn;
                                CSn = isM ? (diag_ok ? LCS2[h] : -1) : viaGap;
                            }
                            CS1[h] = CSn; CI1[h] = CIn; CD1[h] = CDn;
                            if (PH == 2 || k == marker) { if (inband) lds_st<int>(vcur + 4u * (unsigned)(i - vL) + O_CD, CDn); }
                        }
                        Sn[h] = Sv; In[h] = Iv; Dn[h] = Dv;
                        if constexpr (GUESS && PH == 1) {      // the scout's start guess: best cell of diagonal marker-1 / marker
                            if (inband && Sv > -inf) {
                                const unsigned long long key = ((unsigned long long)score_key(Sv) << 32) | (unsigned)i;
                                asm volatile("ds_max_u64 %0, %1" ::"v"(lds_off(&s_team[(k == marker) ? 3 : 2])), "v"(key) : "memory");
                            }
                        }
                        // the running maximum: only when some lane of the half beats it
                        {
                            const unsigned long long gm = __builtin_amdgcn_ballot_w64(Sv > msp);
                            if (gm != 0ull) lds_max_f32_off<O_RED>(__builtin_amdgcn_inverse_ballot_w64(gm) ? vcur : vTrashRed, Sv);
                        }
                        if constexpr (PH == 2) {       // CS of this half's first / last unpruned row, for the pre-test of the convergence test
                            const int firstLane = (vm != 0ull) ? (int)__builtin_ctzll(vm) : -1, lastLane = (vm != 0ull) ? 63 - (int)__builtin_clzll(vm) : -1;
                            const unsigned trashE = vcur + O_TRASH + (unsigned)lane * 16u + 8u;
                            lds_st<int>((lane == firstLane) ? vcur + O_EDGE + 8u * (unsigned)(2 * w + h) : trashE, CS1[h]);
                            lds_st<int>((lane == lastLane) ? vcur + O_EDGE + 8u * (unsigned)(2 * w + h) + 4u : trashE + 4u, CS1[h]);
                        }
                        if constexpr (TB) {                                                        // :548-557
                            const uint32_t nib = (isM ? 0u : (gapIsI ? 1u : 2u)) | (Iptr ? 4u : 0u) | (Dptr ? 8u : 0u);
                            tbacc[h] |= nib << (4 * (k & 7));
                        }
                    }
                    S1 = nuc_f2{Sn[0], Sn[1]}; I1 = nuc_f2{In[0], In[1]}; D1 = nuc_f2{Dn[0], Dn[1]};
                    // the band's ends among this wave's rows (:563-583): first and last surviving row, one lane posts
                    if ((vmAll[0] | vmAll[1]) != 0ull) {
                        const int lowRow = (vmAll[0] != 0ull) ? b + (int)__builtin_ctzll(vmAll[0]) : b + 64 + (int)__builtin_ctzll(vmAll[1]);
                        const int highRow = (vmAll[1] != 0ull) ? b + 127 - (int)__builtin_clzll(vmAll[1]) : b + 63 - (int)__builtin_clzll(vmAll[0]);
                        const unsigned vPost = __builtin_amdgcn_inverse_ballot_w64(1ull) ? vcur : vTrashRed;
                        lds_max_u32_off<O_RED + 4>(vPost, kk16 + (0xFFFFu - (unsigned)lowRow));
                        lds_max_u32_off<O_RED + 8>(vPost, kk16 + (unsigned)highRow);
                    }
                    // mailbox: the upper half's lane 63 is row i-1 of the next block's first row
                    if constexpr (CONV) lds_st<nuc_i4>(vcur + mbRel, nuc_i4{__float_as_int(S1.y), __float_as_int(I1.y), CS1[1], CI1[1]});
                    else lds_st<nuc_i2>(vcur + mbRel, nuc_i2{__float_as_int(S1.y), __float_as_int(I1.y)});
                    LS2 = LS1;
                    if constexpr (CONV) { LCS2[0] = LCS1[0]; LCS2[1] = LCS1[1]; }
                } else { pIn[0] = pIn[1] = 0ull; pLeft[0] = pLeft[1] = 0ull; }
                if (__builtin_expect(b + 127 < Lk, 0)) {    // the block fell out of the band: take the next one
                    while (128 * dblk + 127 < Lk) dblk += W;
                    ring_addr(k);
                    load_q();
                    fresh = 1;
                }
                raA -= 4u;                  // the next diagonal's column is one higher: one slot lower
                if (raA == lds_off(s_ring) - 4u) raA = lds_off(s_ring) + (CAP - 1) * 4u;
                if constexpr (TB) tbPending = true;
                const bool hook = ((k & 7) == 7 || (PH == 1 && k == marker));
                if (hook) {
                    if constexpr (TB) {
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (2 * w))) = tbacc[0];
                        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (2 * w + 1))) = tbacc[1];
                        tbacc[0] = tbacc[1] = 0;
                        tbOff += (unsigned)WINDOW * 4u;
                        tbPending = false;
                    }
                    // every 8th diagonal: stage the next 64 reference columns when the band gets within 8 + 64 columns of them
                    const int need_hi = ((k + 9 - Lk) >> 6) + 1;
                    if (hiBlk < need_hi) { ++hiBlk; if (w == hiBlk % W) load_ring_block(hiBlk); }
                }
                TWL_SETPRIO(0);
                wg_barrier_lds();

                // ---- post: the band of the next diagonal, :563-604 ----
                const nuc_i4 rd = lds_ld<nuc_i4>(vcur + O_RED);
                { const float g = __int_as_float(rd.x); msp = (g > msp) ? g : msp; }
                const int newL = (int)((kk16 + 0xFFFFu) - (unsigned)rd.y);
                const int newU = (int)((unsigned)rd.z - kk16);

                if constexpr (CONV) {                                                              // :585-595
                    if (!converged && k < kEnd - 1) {
                        int conv_S = -1;
                        bool all3 = false;
                        if constexpr (PH == 1) {
                            const int sL = __builtin_amdgcn_readfirstlane(newL), sU = __builtin_amdgcn_readfirstlane(newU);
                            if (k == marker - 1) conv_S = (sL == sU) ? ((3 << 16) | (sL & 0xFFFF)) : -1;
                            else conv_S = (sL == sU) ? (sL & 0xFFFF) : -1;
                        } else {
                            if (threadIdx.x == 0) {
                                int c0 = 0x7fffffff, c1 = (int)0x80000000, c2 = 0;
                                asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
                                lds_st<nuc_i4>(vprev + O_CONV, nuc_i4{c0, c1, c2, c2});
                            }
                            // pre-test (necessary condition): the two end cells of the surviving band hold the same convergence pointer
                            const int cLo = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newL >> 6) % (unsigned)NV));
                            const int cHi = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newU >> 6) % (unsigned)NV) + 4u);
                            const bool maybe = __builtin_amdgcn_ballot_w64(newL <= newU && cLo == cHi) != 0ull;
                            if (maybe) {
                                const unsigned cw = (unsigned)(newU - newL);
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const int i = 128 * dblk + 64 * h + lane;
                                    const bool inr = (unsigned)(i - newL) <= cw;
                                    const unsigned long long rm = __builtin_amdgcn_ballot_w64(inr);
                                    if (rm) {
                                        const int fl = (int)__builtin_ctzll(rm);
                                        const int v = __builtin_amdgcn_readlane(CS1[h], fl);
                                        const bool badS = __builtin_amdgcn_ballot_w64(inr && CS1[h] != v) != 0ull;
                                        const bool badID = __builtin_amdgcn_ballot_w64(inr && (CI1[h] != v || CD1[h] != v)) != 0ull;
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
                        if (all3 && prev_conv_s == conv_S && conv_S != -1) { converged = true; conv_value = prev_conv_s; convf = msp; }
                        prev_conv_s = conv_S;
                    }
                }
                {                                                                                  // :597-604
                    vlo1 = vL; vw1 = vwidth1; vwid1 = (int)vwidth1 + 1;
                    sLo2p = sLo1 + 1; sW2 = sW1; sLo1 = Lk; sW1 = (unsigned)width1;
                    vL = max(max(newL, k + 2 - refLen), 0);
                    vU = min(newU + 1, qLen - 1);
                    Lk = __builtin_amdgcn_readfirstlane(vL);
                    Uk = __builtin_amdgcn_readfirstlane(vU);
                    vcur ^= parx; vprev ^= parx;
                }
                bool ended = false;
                if constexpr (CONV) {
                    if (converged) {                                                               // :607-612
                        if (__builtin_amdgcn_ballot_w64(__int_as_float(max(__float_as_int(msp), 0)) > convf) != 0ull) { conv_logic = true; go = false; ended = true; }
                    }
                }
                if (!ended) {
                    ++k;
                    if (__builtin_expect((unsigned)(Uk - Lk) >= (unsigned)fcap, 0)) {
                        if (k >= kEnd) {}
                        else if (Lk > Uk) { tile_err = 1; go = false; }
                        else if (Uk - Lk + 1 > fLen) { tile_err = 2; go = false; }
                        else if ((Uk >> 7) - (Lk >> 7) >= W) { tile_err = kErrOverflow; go = false; }   // it really outgrew this window
                        else if (128 * dblk + 127 < Lk) {
                            // a band this wide can need the successor of a block on the diagonal the block leaves it: advance before the activity test
                            while (128 * dblk + 127 < Lk) dblk += W;
                            ring_addr(k); load_q(); fresh = 1;
                        }
                    }
                }
            };

            if (steps_left < 0) { tile_err = 3; go = false; }
            {
                using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
                const int kA = min(kEnd, marker - 1);
                while (go && k < kA) step(T0{});
                const int kB = min(kEnd, marker + 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) { CS1[h] = -1; CI1[h] = kIB; CD1[h] = kDB; LCS2[h] = -1; }
                while (go && k < kB) step(T1{});
                if constexpr (MT == 2) {       // a scout ends here: its path is traced back from the better of the best cells of its two marker diagonals
                    if (go && k == marker + 1 && k < kEnd) {
                        const unsigned long long b0 = s_team[2], b1 = s_team[3];
                        const unsigned k0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(b0 >> 32)), k1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(b1 >> 32));
                        const int i0 = __builtin_amdgcn_readfirstlane((int)(b0 & 0xFFFFFFFFu)), i1 = __builtin_amdgcn_readfirstlane((int)(b1 & 0xFFFFFFFFu));
                        if ((k0 | k1) != 0u) { conv_value = (k1 >= k0) ? (i1 & 0xFFFF) : ((3 << 16) | (i0 & 0xFFFF)); conv_logic = true; }
                        else tile_err = 1;
                        go = false;
                    }
                }
                const int kCap = min(kEnd, 65534);      // k + 1 must fit the 16-bit tag
                while (go && k < kCap) step(T2{});
                if (go && k < kEnd) { tile_err = kErrOverflow; go = false; }
            }
            const int last_k = conv_logic ? k : k - 1;
            steps_left -= (long long)(last_k + 1);
            const unsigned tile_cells = (unsigned)__builtin_amdgcn_readfirstlane((int)vcells);
            const int lo1 = __builtin_amdgcn_readfirstlane(vlo1);

            cells += tile_cells;
            dbg_lastk = last_k; dbg_conv = conv_value; dbg_L = Lk; dbg_U = Uk;
            if (tile_err != 0) { err = tile_err; break; }
            if (!denomOne) {
                if (guardBad) s_misc[4] = 1;
                __syncthreads();
                guardBad = __builtin_amdgcn_readfirstlane(s_misc[4]) != 0;
                if (guardBad) { err = kErrGuard; break; }
            }
            if (tbPending) {
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (2 * w))) = tbacc[0];
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (2 * w + 1))) = tbacc[1];
            }

            // ---- tile exit, :615-682 ----
            int conv_q = 0, conv_r = 0, tb_state = 0, start_k = 0;
            bool bad = false;
            if (!conv_logic && last_k >= marker) {                            // :633-635 needs CS[last_k][0]
                const int Llast = lo1;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int bh = 128 * dblk + 64 * h;
                    if (Llast >= bh && Llast <= bh + 63 && lane == Llast - bh) s_misc[1] = CS1[h];
                }
                __syncthreads();
                conv_value = __builtin_amdgcn_readfirstlane(s_misc[1]);
            }
            if (conv_logic || last_k >= marker) {
                conv_q = conv_value & 0xFFFF;
                tb_state = (conv_value >> 16) & 0xFFFF;
                if (tb_state > 3) bad = true;
                else {
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
                {   // Traceback, :134-231: one lane walks the pointers out of LDS patches of 64 rows x 16 groups of 8 diagonals (the ring is dead until the next tile)
                    constexpr int PG = 16;
                    static_assert(sizeof(s_ring) >= PG * 64 * sizeof(uint32_t), "traceback patch lives in the ring");
                    uint32_t *s_patch = reinterpret_cast<uint32_t *>(s_ring);
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
                if constexpr (MT == 1) {          // the segment and the record of this tile (read by the stitch launch)
                    if (cnt <= a.mt_segcap) {
                        for (int t = lane; t < cnt; t += 64) out[t] = s_rev[n - 1 - skip - t];
                        if (lane == 0) {
                            int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + slot) * kMtRec;
                            rc[1] = jobRef; rc[2] = jobQry; rc[3] = ref_idx; rc[4] = qry_idx; rc[5] = last_tile ? 1 : 0; rc[6] = cnt;
                            rc[7] = tailDir; rc[8] = tailLen; rc[9] = (int32_t)tile_cells; rc[0] = 1;
                        }
                    }
                } else if constexpr (MT == 2) {   // where the scout's path crosses the anti-diagonals of its tile boundary
                    if (lane == 0) {
                        int32_t *sp = a.mt_spath + (size_t)mtx * (size_t)a.mt_sp_pitch;
                        const int dMax = min(spHi, R + Q - 2);
                        for (int d = spLo; d <= dMax; ++d) sp[d] = -1;
                        int r = jobRef, q = jobQry;
                        for (int t = n - 2; t >= 0; --t) {
                            const int dir = s_rev[t];
                            r += (dir != 1) ? 1 : 0; q += (dir != 2) ? 1 : 0;
                            const int d = r + q;
                            if (d >= spLo && d <= dMax) sp[d] = q;
                        }
                    }
                } else
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
            if constexpr (MT == 1 || MT == 2) last_tile = true;      // one tile per job
        }

        __syncthreads();
        if constexpr (MT == 2) {
            if (threadIdx.x == 0 && err != 0 && a.mt_stat) atomicAdd(&a.mt_stat[2], 1ull);
        }
        if constexpr (MT == 1) {          // a tile job that ended with an error code leaves that as its record (read by the stitch launch)
            if (threadIdx.x == 0 && err != 0 && jobRef >= 0) {
                int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + slot) * kMtRec;
                rc[1] = jobRef; rc[2] = jobQry; rc[9] = (int32_t)(unsigned)cells; rc[10] = err; rc[11] = WINDOW; rc[0] = 2;
            }
        }
        if (threadIdx.x == 0 && MT == 0) {
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

// twilight_amd/csrc/talco_kernel.hip.h -- gfx950 device code of the tiled TALCO-XDrop level-batch aligner.
//
// What it computes is exactly Talco_xdrop::Align_freq / Tile / Traceback of the reference CPU path
// (/root/reference/src/TALCO-XDrop.cpp:62-108, :233-689, :134-231) in fp32 with the x86 TALCO_SIMD
// operation order; how it computes it is MI355X-specific and shares no structure with the
// reference's own GPU kernels (src/hip/device-function.hip.cpp -- int16 scores, marker 200, a
// different algorithm variant; not the oracle, not consulted for this design).
//
// Mapping:
//   * one persistent workgroup of W waves aligns one pair at a time, pulled from a device queue;
//   * lane <-> query row i ("systolic"): a 64-row block of the query profile lives in the lanes of a
//     virtual wave for as long as the band [L,U] touches it; the products q[m]*M[l][m] (the first
//     rounding of the reference's (q*M)*r) are computed once per row and kept in 25 VGPRs;
//   * the reference column r[j], j = k - i, streams past the lanes: it is staged once per 64 columns
//     into an LDS ring and read conflict-free (consecutive lanes -> consecutive 32-byte columns);
//   * H/I of row i-1 arrive by one DPP wave_shr:1 per value (lane 0 patched from an LDS mailbox),
//     H/D of row i stay in the lane's own registers: no DP row ever goes to LDS or HBM;
//   * per anti-diagonal the workgroup reduces {running max, first/last unpruned row} with three LDS
//     atomics and ONE barrier; convergence pointers (k > marker) add a second barrier;
//   * traceback pointers (4 bit/cell, k <= marker) are packed 8 diagonals per dword per lane and
//     written coalesced to an HBM scratch tile [k/8][row mod window]; the tile's path is walked by
//     one lane and streamed reversed into the output.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace twl {

constexpr int kIB = -2;   // I_BOUNDARY, TALCO-XDrop.cpp:33
constexpr int kDB = -3;   // D_BOUNDARY, TALCO-XDrop.cpp:34
constexpr int kMaxMarker = 1024;
constexpr int kErrOverflow = -1;   // internal: band outgrew this kernel's row window -> relaunch wide
constexpr int kErrGuard = -2;      // internal: an operand outside the hoisted-reciprocal division's range -> relaunch on the IEEE-division kernel

struct KArgs {
    const float *cols;        // packed columns [pair][2][seq_len][P+2]: f0..f(P-1), gapOpen, gapExtend (32 B for P=6, 96 B for P=22)
    const int32_t *len;       // [pair][2]
    const int32_t *num;       // [pair][2]
    int8_t *aln;              // [pair][2*seq_len]
    int32_t *aln_len;         // [pair]
    int16_t *err;             // [pair]
    unsigned long long *cells;// [pair]
    uint32_t *tb;             // [grid][tb_words]
    int32_t *queue;           // [1] next work item
    const int32_t *items;     // [n_items] pair ids in launch order (cost-sorted)
    int32_t n_items;
    int32_t seq_len;
    int32_t tb_words;         // per workgroup
    int32_t *dbg;             // optional [pair][16] debug record (nullptr = off)
    int32_t n_pairs_total;    // pairs of the batch (the stamp build writes its record after the per-pair debug records)
    int32_t *hb;              // optional host-mapped heartbeat [16] written by workgroup 0 (debug only)
    int32_t step_slack;       // watchdog: a pair may run at most 32*(R+Q) + step_slack diagonals in total
    float gap_open, gap_extend, gap_char;
    const uint8_t *gc_zero;   // optional [pair]: 1 = this pair's gapCharScore is 0 whatever gap_char says (alignment-cpu.cpp:88 decides per pair)
    int32_t xdrop, flen, marker;
    float M[441];             // scoreMatrix[l][m] row-major, (P-1)x(P-1): 5x5 or 21x21
    // matrix mode 4 (protein, few pairs): column scores precomputed by score_matrix_kernel, diagonal-major per pair:
    // sim[sim_off[pair] + (i + j) * pitch + i] with i = query row, j = reference column, pitch = (Q + 63) & ~63
    const float *sim;
    const long long *sim_off;
};

#ifdef TWL_KERNEL_STAMPS
#define TWL_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define TWL_STAMP(var)
#endif

#ifdef TWL_KERNEL_DEBUG
__device__ __forceinline__ void heartbeat(const KArgs &a, int slot, int v)
{
    if (a.hb && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&a.hb[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#else
__device__ __forceinline__ void heartbeat(const KArgs &, int, int) {}
#endif

// Single-lane LDS atomics issued directly: hipcc otherwise wraps every LDS atomic in its wave-reduction prologue
// (v_mbcnt/readfirstlane) even when the call site is already restricted to one lane.
__device__ __forceinline__ unsigned lds_off(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_max_i32(int *p, int v) { asm volatile("ds_max_i32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_min_i32(int *p, int v) { asm volatile("ds_min_i32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_or_b32(int *p, int v) { asm volatile("ds_or_b32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory"); }

__device__ __forceinline__ float dpp_shr1_f(float from_prev_wave, float src)
{
    // lane t <- src[t-1]; lane 0 keeps `from_prev_wave` (DPP wave_shr:1, bound_ctrl=0)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(from_prev_wave), __float_as_int(src), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int dpp_shr1_i(int from_prev_wave, int src)
{
    return __builtin_amdgcn_update_dpp(from_prev_wave, src, 0x138, 0xf, 0xf, false);
}

// max over the 64 lanes (all lanes must be active); result broadcast as a wave-uniform value.
__device__ __forceinline__ float wave_max_f32(float x)
{
#define TWL_DPP_MAX(ctrl, rmask, bmask) \
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, rmask, bmask, false)))
    TWL_DPP_MAX(0x111, 0xf, 0xf);   // row_shr:1
    TWL_DPP_MAX(0x112, 0xf, 0xf);   // row_shr:2
    TWL_DPP_MAX(0x114, 0xf, 0xf);   // row_shr:4
    TWL_DPP_MAX(0x118, 0xf, 0xf);   // row_shr:8
    TWL_DPP_MAX(0x142, 0xa, 0xf);   // row_bcast:15
    TWL_DPP_MAX(0x143, 0xc, 0xf);   // row_bcast:31
#undef TWL_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// order-preserving float <-> int key for LDS integer atomic max
__device__ __forceinline__ int f2key(float f)
{
    int b = __float_as_int(f);
    if (b == (int)0x80000000) b = 0;          // -0.0 == +0.0
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k)
{
    return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff));
}

template <int P, int W, int RPL, bool PRE, bool REFLDS, bool QREG = true>
struct Cfg {
    static_assert(P == 6 || P == 22, "profile width");
    static_assert(!(PRE && P != 6), "precomputed q*M rows exist for P=6 only");
    static constexpr int CW = P + 2;            // floats per packed column
    static constexpr int F4 = CW / 4;           // float4 per packed column
    static constexpr int NV = W * RPL;          // virtual waves = 64-row blocks resident at once
    static constexpr int WINDOW = 64 * NV;      // rows
    static constexpr int NB = NV + 2;           // ring blocks
    static constexpr int CAP = 64 * NB;         // ring columns
    static constexpr int RING_F4 = REFLDS ? CAP * F4 : F4;
    static constexpr int THREADS = 64 * W;
};

// MM = matrix mode of the nucleotide path (chosen by the host from the matrix values; every mode computes the same sums):
//   0  general 5x5;
//   1  N row and column all zero (scoring-matrix.cpp:104 without -w): their 9 products are +-0 and adding +-0 never changes a
//      non-zero partial sum, so the 4x4 core is enough (this can only flip the sign of a zero);
//   2  mode 1 and M[l][m] = A (l==m), B (|l-m|==2, transitions), C (otherwise): q[m]*M[l][m] takes 3 values per m -> 12 first
//      products instead of 16 and 3 live scalars instead of 16.
template <int P, int W, int RPL, bool PRE, bool REFLDS, bool QREG = true, int MINW = 1, int MM = 0>
__global__ __launch_bounds__(64 * W, MINW) void talco_kernel(KArgs a)
{
    using C = Cfg<P, W, RPL, PRE, REFLDS, QREG>;
    constexpr int NV = C::NV, WINDOW = C::WINDOW, NB = C::NB, CAP = C::CAP, CW = C::CW, F4 = C::F4;
    constexpr int MS = P - 1;                    // matrix side

    // Protein sparse mode (MM == 3): the score loop runs over the NON-ZERO letters of the reference column only.  A skipped
    // letter l has r[l] == 0, so each of its 21 products is +-0 and so is their block sum; adding +-0 to the running sum (which
    // starts at +0 and therefore is never -0) changes nothing, so the result is bit-identical to the dense loop.  The letters are
    // visited in ascending order, the reference's order.  s_rmask holds the non-zero-letter bitmask of every ring column (computed
    // once when the column enters the ring), s_M the matrix rows padded to 24 floats.
    constexpr bool SPARSE = (P == 22) && (MM == 3);
    constexpr bool PRESIM = (P == 22) && (MM == 4);      // scores come from score_matrix_kernel (same arithmetic, done ahead by the whole GPU)
    static_assert(!PRESIM || REFLDS, "presim mode takes the gap penalties of the reference column from the LDS ring");
    static_assert(!SPARSE || REFLDS, "sparse mode reads the reference column from the LDS ring");
    __shared__ float4 s_ring[C::RING_F4];
    __shared__ uint32_t s_rmask[SPARSE ? CAP : 1];
    __shared__ float4 s_M4[SPARSE ? 21 * 6 : 1];
    __shared__ int4 s_exch[2][NV];
    __shared__ int s_red[3][4];      // {max key, first unpruned row (min), last unpruned row (max), -}
    __shared__ int s_conv[2][4];     // {vmin, vmax, flags, -}
    __shared__ int s_misc[4];
    // Offset-addressed mirror of the reference's two rotating CD rows (TALCO-XDrop.cpp:279,294,308).  The
    // reference reads CD[(k+1)%2][offsetUp] unguarded above (:535): for the new top cell of a growing band
    // offsetUp == width(k-1), i.e. whatever an older, wider diagonal left there -- and that value takes part in
    // the convergence test (:587).  Only this row's addressing is observable, so only it is mirrored.
    __shared__ int s_cd[2][WINDOW + 1];
    __shared__ int8_t s_rev[2 * kMaxMarker + 16];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tb = a.tb + (size_t)blockIdx.x * (size_t)a.tb_words;
    const float gapOpen = a.gap_open, gapExtend = a.gap_extend;
    const int marker = a.marker;
    const float inf = (float)(2.0 * (double)a.xdrop + 1.0);     // TALCO-XDrop.cpp:252
    const float xdropf = (float)a.xdrop;

    if constexpr (SPARSE) {
        float *sm = reinterpret_cast<float *>(s_M4);
        for (int t = threadIdx.x; t < 21 * 24; t += C::THREADS) { const int l = t / 24, m = t % 24; sm[t] = (m < 21) ? a.M[21 * l + m] : 0.0f; }
    }
    heartbeat(a, 0, 1);
    for (;;) {
        if (threadIdx.x == 0) s_misc[0] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(s_misc[0]);
        heartbeat(a, 1, item);
        if (item >= a.n_items) break;
        const int pair = __builtin_amdgcn_readfirstlane(a.items[item]);
        const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
        const float gc = (a.gc_zero && a.gc_zero[pair]) ? 0.0f : a.gap_char;
        const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];   // :255,:269
        const bool denomOne = (denom == 1.0f);
        const float4 *colsR = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * CW);
        const float4 *colsQ = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * CW);
        int8_t *out = a.aln + (size_t)pair * 2 * (size_t)a.seq_len;
        const float *simP = PRESIM ? a.sim + a.sim_off[pair] : nullptr;
        const int simPitch = (Q + 63) & ~63;

        // An empty side produces no path (the caller emits the all-gap path, alignment-cpu.cpp:89-90).
        // NOTE on control flow: no `continue`, and every single-lane block is followed by a workgroup
        // barrier before a loop back-edge.  With a divergent block next to the latch LLVM splits the
        // back-edge and lets the other lanes run ahead into the next iteration's barrier (observed: the
        // workgroup re-read a stale work item forever).
        int ref_idx = 0, qry_idx = 0, tile = 0, pos = 0, err = 0;
        bool last_tile = (R <= 0 || Q <= 0);
        unsigned long long cells = 0;
        // watchdog: every loop below is bounded by it.  A tile advances at least marker-1 cells and runs at most R+Q diagonals.
        int steps_left = (int)min((long long)(R + Q + 2) * ((R + Q) / (max(a.marker, 2) - 1) + 4) + a.step_slack, 0x7fffffffll);
        int dbg_lastk = 0, dbg_conv = 0, dbg_L = 0, dbg_U = 0;
#ifdef TWL_KERNEL_STAMPS
        unsigned long long st_slots = 0, st_bar = 0, st_post = 0, st_n = 0, st_act = 0, st_exit = 0, st_setup = 0;
        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

        while (!last_tile) {   // ---- Align_freq tile loop, TALCO-XDrop.cpp:77-106 ----
            TWL_STAMP(t_tile0);
            int refLen = R - ref_idx, qLen = Q - qry_idx;
            const int simK0 = ref_idx + qry_idx;                                      // global anti-diagonal of the tile's first cell
            const int fLen = min(a.flen, min(refLen, qLen));                          // :258
            // ---- per-slot (virtual wave) state ----
            float S1[RPL], I1[RPL], D1[RPL], LS2[RPL];
            int CS1[RPL], CI1[RPL], CD1[RPL], LCS2[RPL];
            float qv[RPL][QREG ? P : 1], gopq[RPL], gexq[RPL];
            float qM[RPL][PRE ? 25 : 1];
            int blk[RPL], uph[RPL];
            uint32_t tbacc[RPL];
            bool tbdirty[RPL], q5any[RPL];

            auto load_col = [&](const float4 *base, size_t col, bool ok, float *dst /*[CW]*/) {
#pragma unroll
                for (int t = 0; t < F4; ++t) {
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) c = base[F4 * col + t];
                    dst[4 * t] = c.x; dst[4 * t + 1] = c.y; dst[4 * t + 2] = c.z; dst[4 * t + 3] = c.w;
                }
            };
            auto load_q = [&](int r) {
                if constexpr (!QREG) { q5any[r] = true; gopq[r] = gexq[r] = 0.f; return; }     // columns are re-read per cell
                const int i = 64 * blk[r] + lane;
                float cb[CW];
                load_col(colsQ, (size_t)(qry_idx + i), qry_idx + i < Q, cb);
#pragma unroll
                for (int t = 0; t < P; ++t) qv[r][QREG ? t : 0] = cb[t];
                gopq[r] = cb[P]; gexq[r] = cb[P + 1];
                if constexpr (PRE) {
#pragma unroll
                    for (int l = 0; l < 5; ++l)
#pragma unroll
                        for (int m = 0; m < 5; ++m) qM[r][5 * l + m] = cb[m] * a.M[5 * l + m];   // first rounding of (q*M)*r, :386
                }
                q5any[r] = __builtin_amdgcn_ballot_w64(cb[P - 1] != 0.0f) != 0ull;
            };
            auto load_ring_block = [&](int B) {
                if constexpr (REFLDS) {
                    const int col = 64 * B + lane;
                    const int slot = (B % NB) * 64 + lane;
                    uint32_t mk = 0;
#pragma unroll
                    for (int t = 0; t < F4; ++t) {
                        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (col < refLen) c = colsR[F4 * (size_t)(ref_idx + col) + t];
                        s_ring[t * CAP + slot] = c;
                        if constexpr (SPARSE) {
                            if (4 * t + 0 < 21) mk |= (c.x != 0.0f) ? (1u << (4 * t + 0)) : 0u;
                            if (4 * t + 1 < 21) mk |= (c.y != 0.0f) ? (1u << (4 * t + 1)) : 0u;
                            if (4 * t + 2 < 21) mk |= (c.z != 0.0f) ? (1u << (4 * t + 2)) : 0u;
                            if (4 * t + 3 < 21) mk |= (c.w != 0.0f) ? (1u << (4 * t + 3)) : 0u;
                        }
                    }
                    if constexpr (SPARSE) s_rmask[slot] = mk;
                }
            };

#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int vw = r * W + w;
                blk[r] = vw;
                uph[r] = (vw == 0) ? 0 : CAP - 64 * vw;       // (k - 64*blk) mod CAP at k = 0
                tbacc[r] = 0; tbdirty[r] = false;
                S1[r] = I1[r] = D1[r] = LS2[r] = -1.0f;       // never read before written for in-band cells
                CS1[r] = -1; CI1[r] = kIB; CD1[r] = kDB; LCS2[r] = -1;
                load_q(r);
            }
            int hiBlk = 1;
            if (w == 0 % W) load_ring_block(0);
            if (w == 1 % W) load_ring_block(1);
            for (int t = threadIdx.x; t < 2 * (WINDOW + 1); t += C::THREADS) (&s_cd[0][0])[t] = kDB;      // :308
            if (threadIdx.x == 0) {
                s_red[0][0] = f2key(-inf); s_red[0][1] = 0x7fffffff; s_red[0][2] = -1;
                s_conv[0][0] = 0x7fffffff; s_conv[0][1] = (int)0x80000000; s_conv[0][2] = 0;
                s_conv[1][0] = 0x7fffffff; s_conv[1][1] = (int)0x80000000; s_conv[1][2] = 0;
            }
            __syncthreads();

            // ---- Tile, TALCO-XDrop.cpp:233-689 ----
            int Lk = 0, Uk = 0, L1 = 2, U1 = -2, L2 = 1, U2 = -1;      // :296-297 (rows k, k-1, k-2)
            // running maximum / X-drop reference / score at convergence, kept as order-preserving integer keys so that
            // their bookkeeping is scalar integer work (:259-260,:501-503,:607-612)
            int mspKey = f2key(-inf), msKey = 0, convKey = 0;
            bool converged = false, conv_logic = false;
            int conv_value = 0, prev_conv_s = -1, last_k = 0;
            const int kEnd = refLen + qLen - 1;
            int k = 0;
            int tile_err = 0;
            int rs3 = 0;                                 // k % 3, kept incrementally
            unsigned tile_cells = 0;                     // < 2^32 per tile: at most (refLen+qLen) diagonals x 4096 cells

            heartbeat(a, 2, tile);
#ifdef TWL_KERNEL_STAMPS
            st_setup += __builtin_amdgcn_s_memtime() - t_tile0;
#endif
            for (; k < kEnd; ++k) {
                heartbeat(a, 3, k);
                TWL_STAMP(t_head);
                // one rarely-taken exit for all four stop conditions (decoded after the loop)
                if (__builtin_expect((--steps_left < 0) | (Lk > Uk) | (Uk - Lk + 1 > fLen) | ((Uk >> 6) - (Lk >> 6) >= NV), 0)) {
                    tile_err = (steps_left < 0) ? 3 : (Lk > Uk) ? 1                       // :323-329 band emptied by X-drop
                             : (Uk - Lk + 1 > fLen) ? 2 : kErrOverflow;                    // :331-338 wider than fLen / than this window
                    break;
                }
                tile_cells += (unsigned)(Uk - Lk + 1);
                const int par = k & 1;
                const float msp = key2f(mspKey);
                const float thr = key2f(msKey) - xdropf;                   // :495
                const bool pb = (k >= marker - 1);
                const bool edgeStep = (tile == 0) && (Lk == 0 || Uk == k);
                const bool special = (k == 0) | edgeStep;                  // first diagonal of a tile / first row or column of tile 0
                // the slot one past diagonal k-1 in the mirrored CD row (see s_cd): one broadcast read per diagonal
                const int staleCD = pb ? s_cd[par ^ 1][(U1 >= L1) ? (U1 - L1 + 1) : 0] : kDB;
                // bands as (low, width) with an empty band mapped to an unreachable low: one unsigned compare per test
                const unsigned wk = (unsigned)(Uk - Lk);
                const int L1e = (U1 >= L1) ? L1 : 0x3fffffff;
                const unsigned w1 = (U1 >= L1) ? (unsigned)(U1 - L1) : 0u;
                const int L2p = (U2 >= L2) ? L2 + 1 : 0x3fffffff;
                const unsigned w2 = (U2 >= L2) ? (unsigned)(U2 - L2) : 0u;

                if constexpr (REFLDS) {
                    const int need_hi = ((k - Lk) >> 6) + 1;
                    if (hiBlk < need_hi) { ++hiBlk; if (w == hiBlk % W) load_ring_block(hiBlk); }
                }
                if (threadIdx.x == 0) {   // reset the reduction slot of the next diagonal
                    const int nx = (rs3 == 2) ? 0 : rs3 + 1;
                    s_red[nx][0] = f2key(-inf); s_red[nx][1] = 0x7fffffff; s_red[nx][2] = -1;
                }

#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int vw = r * W + w;
                    if (64 * blk[r] + 63 < Lk) {               // block fell out of the band: take the next one
                        while (64 * blk[r] + 63 < Lk) { blk[r] += NV; uph[r] -= WINDOW; if (uph[r] < 0) uph[r] += CAP; }
                        load_q(r);
                    }
                    const int b = 64 * blk[r];
                    // The block takes part when the band touches it or will reach it on the next diagonal (its lane 0 then needs
                    // S[k-1][b-1] now, to have S[k-2][i-1] next time).  In that one extra step every lane is out of band.
                    const bool act = (b <= Uk + 1) && (b + 63 >= Lk);
                    if (act) {
                        const int4 e = s_exch[par ^ 1][(vw + NV - 1) % NV];
                        const float LS1 = dpp_shr1_f(__int_as_float(e.x), S1[r]);
                        const float LI1 = dpp_shr1_f(__int_as_float(e.y), I1[r]);
                        const int LCS1 = dpp_shr1_i(e.z, CS1[r]);
                        const int LCI1 = dpp_shr1_i(e.w, CI1[r]);
                        const int i = b + lane;
                        const int j = k - i;
                        const bool inband = (unsigned)(i - Lk) <= wk;
                        // reference column r[j] (and, when it is not register-resident, the query column q[i])
                        float rc[CW];
                        int rs = 0;
                        uint32_t rmask = 0;
                        float presim = 0.0f;
                        if constexpr (PRESIM) {
                            if (inband) presim = simP[(size_t)(k + simK0) * (size_t)simPitch + (size_t)(qry_idx + i)];
                            rs = uph[r] - lane; rs += (rs < 0) ? CAP : 0;
                            const float4 c = s_ring[(F4 - 1) * CAP + rs];          // {X letter, gap, gapOpen, gapExtend}
                            rc[CW - 4] = c.x; rc[CW - 3] = c.y; rc[CW - 2] = c.z; rc[CW - 1] = c.w;
                        } else if constexpr (SPARSE) {
                            rs = uph[r] - lane; rs += (rs < 0) ? CAP : 0;
                            const float4 c = s_ring[(F4 - 1) * CAP + rs];          // {X letter, gap, gapOpen, gapExtend}
                            rc[CW - 4] = c.x; rc[CW - 3] = c.y; rc[CW - 2] = c.z; rc[CW - 1] = c.w;
                            rmask = inband ? s_rmask[rs] : 0u;
                        } else if constexpr (REFLDS) {
                            rs = uph[r] - lane; rs += (rs < 0) ? CAP : 0;
#pragma unroll
                            for (int t = 0; t < F4; ++t) {
                                const float4 c = s_ring[t * CAP + rs];
                                rc[4 * t] = c.x; rc[4 * t + 1] = c.y; rc[4 * t + 2] = c.z; rc[4 * t + 3] = c.w;
                            }
                        } else {
                            load_col(colsR, (size_t)(ref_idx + j), inband, rc);
                        }
                        float qcb[QREG ? 1 : CW];
                        if constexpr (!QREG) { load_col(colsQ, (size_t)(qry_idx + i), inband, qcb); gopq[r] = qcb[P]; gexq[r] = qcb[P + 1]; }
                        const float *q = QREG ? qv[r] : qcb;
                        const float rg = rc[P - 1], gopr = rc[P], gexr = rc[P + 1];
                        const bool rgAny = __builtin_amdgcn_ballot_w64(inband && rg != 0.0f) != 0ull;
                        float numer = 0.0f;
                        if constexpr (PRESIM) {
                            // nothing to compute: see below
                        } else if constexpr (P == 6 && MM == 2 && !PRE) {
                            const float mA = a.M[0], mB = a.M[2], mC = a.M[1];
                            float pa[4], pb[4], pc[4];
#pragma unroll
                            for (int m = 0; m < 4; ++m) { pa[m] = q[m] * mA; pb[m] = q[m] * mB; pc[m] = q[m] * mC; }
#pragma unroll
                            for (int l = 0; l < 4; ++l) {
                                float t[4];
#pragma unroll
                                for (int m = 0; m < 4; ++m) t[m] = ((l == m) ? pa[m] : (((l ^ m) == 2) ? pb[m] : pc[m])) * rc[l];
                                const float sl = ((t[0] + t[1]) + t[2]) + t[3];
                                numer = (l == 0) ? sl : numer + sl;
                            }
                        } else if constexpr (P == 6 && MM >= 1) {
#pragma unroll
                            for (int l = 0; l < 4; ++l) {
                                float t[4];
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    if constexpr (PRE) t[m] = qM[r][5 * l + m] * rc[l];
                                    else t[m] = (q[m] * a.M[5 * l + m]) * rc[l];
                                }
                                const float sl = ((t[0] + t[1]) + t[2]) + t[3];
                                numer = (l == 0) ? sl : numer + sl;
                            }
                        } else if constexpr (P == 6) {
                            // column score, :378-395 (order: (((t0+t1)+t2)+t3)+t4 per l, accumulated over l)
#pragma unroll
                            for (int l = 0; l < 5; ++l) {
                                float t[5];
#pragma unroll
                                for (int m = 0; m < 5; ++m) {
                                    if constexpr (PRE) t[m] = qM[r][5 * l + m] * rc[l];
                                    else t[m] = (q[m] * a.M[5 * l + m]) * rc[l];
                                }
                                const float sl = (((t[0] + t[1]) + t[2]) + t[3]) + t[4];
                                numer = (l == 0) ? sl : numer + sl;
                            }
                        } else if constexpr (SPARSE) {
                            // protein column score over the non-zero letters of r only (see SPARSE above); same per-letter order as below
                            const float *ringf = reinterpret_cast<const float *>(s_ring);
                            uint32_t mk = rmask;
                            while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                                if (mk != 0u) {
                                    const int l = __builtin_ctz(mk);
                                    mk &= mk - 1u;
                                    const float rl = ringf[(size_t)((l >> 2) * CAP + rs) * 4 + (l & 3)];
                                    float Mr[24];
#pragma unroll
                                    for (int t = 0; t < 6; ++t) {
                                        const float4 c = s_M4[l * 6 + t];
                                        Mr[4 * t] = c.x; Mr[4 * t + 1] = c.y; Mr[4 * t + 2] = c.z; Mr[4 * t + 3] = c.w;
                                    }
#pragma unroll
                                    for (int m = 16; m < 21; ++m) numer += (rl * q[m]) * Mr[m];
                                    float v[8];
#pragma unroll
                                    for (int t = 0; t < 8; ++t) v[t] = (q[t] * Mr[t]) * rl + (q[8 + t] * Mr[8 + t]) * rl;
                                    numer += ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
                                }
                            }
                        } else {
                            // protein column score, :409-430: per l the scalar tail m=16..20 first, then the two 8-lane
                            // blocks v[t] = (q[t]*M[l][t])*r[l] + (q[8+t]*M[l][8+t])*r[l] summed left to right
#pragma unroll
                            for (int l = 0; l < 21; ++l) {
                                const float rl = rc[l];
#pragma unroll
                                for (int m = 16; m < 21; ++m) numer += (rl * q[m]) * a.M[21 * l + m];
                                float v[8];
#pragma unroll
                                for (int t = 0; t < 8; ++t) v[t] = (q[t] * a.M[21 * l + t]) * rl + (q[8 + t] * a.M[21 * l + 8 + t]) * rl;
                                numer += ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
                            }
                        }
                        if (!PRESIM && q5any[r]) {
                            if constexpr (SPARSE) {      // (r[l]*q[gap])*gc is +-0 for the skipped letters
                                const float *ringf = reinterpret_cast<const float *>(s_ring);
                                uint32_t mk = rmask;
                                while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                                    if (mk != 0u) {
                                        const int l = __builtin_ctz(mk);
                                        mk &= mk - 1u;
                                        numer += (ringf[(size_t)((l >> 2) * CAP + rs) * 4 + (l & 3)] * q[P - 1]) * gc;
                                    }
                                }
                            } else {
#pragma unroll
                                for (int l = 0; l < MS; ++l) numer += (rc[l] * q[P - 1]) * gc;   // :394 / :432
                            }
                        }
                        if (!PRESIM && rgAny) {
#pragma unroll
                            for (int m = 0; m < MS; ++m) numer += (rg * q[m]) * gc;              // :395 / :433
                        }
                        const float sim = PRESIM ? presim : (denomOne ? numer : numer / denom);  // :444

                        const unsigned t1 = (unsigned)(i - L1e);
                        const bool up_ok = t1 <= w1;                      // i   in band(k-1)
                        const bool left_ok = (t1 - 1u) <= w1;             // i-1 in band(k-1)
                        const bool diag_ok = (unsigned)(i - L2p) <= w2;   // i-1 in band(k-2)
                        float match = diag_ok ? LS2[r] + sim : -inf;                            // :445-450
                        if (special) {
                            if (k == 0) match = sim;
                            else if (i == 0 || j == 0) {
                                int far = max(i, j) - 1; far = far < 0 ? 0 : far;
                                match = (sim + gapOpen) + gapExtend * (float)far;
                            }
                        }
                        const float delOp = up_ok ? S1[r] + gopr : -inf;                          // :456-463
                        const float delExt = up_ok ? D1[r] + gexr : -inf;
                        const float insOp = left_ok ? LS1 + gopq[r] : -inf;
                        const float insExt = left_ok ? LI1 + gexq[r] : -inf;
                        const bool Iptr = insExt >= insOp;                                        // :468-475
                        const bool Dptr = delExt >= delOp;
                        const float Iv = Iptr ? insExt : insOp;
                        const float Dv = Dptr ? delExt : delOp;
                        // :477-494 without combining lane masks: G = (I > D ? I : D) is the better gap state (D wins ties),
                        // M wins when match >= G (match >= I and match >= D); otherwise the state is I iff I > D
                        const bool gapIsI = Iv > Dv;
                        const float Gv = gapIsI ? Iv : Dv;
                        const bool isM = match >= Gv;
                        float Sv = isM ? match : Gv;
                        Sv = (Sv < thr) ? -inf : Sv;                                            // :495-497

                        // Out-of-band lanes compute too; their registers are never consumed (every reader tests the stored
                        // band of the producing diagonal first), so only LDS/HBM side effects are guarded.
                        if (pb) {                                                               // :520-547
                            // propagation along the chosen predecessors (k > marker) ...
                            const int viaS = (LCS1 != -1) ? LCS1 : kIB;
                            int CIn = left_ok ? (Iptr ? LCI1 : viaS) : kIB;
                            // :534-538; offsetUp >= 0 always holds; above the stored band the reference reads the stale slot
                            const int cdUp = up_ok ? CD1[r] : staleCD;
                            int CDn = Dptr ? cdUp : ((CS1[r] != -1) ? CS1[r] : kDB);
                            const int viaGap = gapIsI ? CIn : CDn;
                            // M without a diagonal predecessor (pruned edge cells; first row/column of tile 0 at the marker): the reference
                            // reads outside its row there (:541); defined as "unset", as in the oracle
                            int CSn = isM ? (diag_ok ? LCS2[r] : -1) : viaGap;
                            if (__builtin_expect(k <= marker, 0)) {          // ... and the two diagonals where the markers are planted
                                const int i16 = i & 0xFFFF;
                                if (k == marker) { CSn = i16; CIn = (1 << 16) | i16; CDn = (2 << 16) | i16; }
                                else { CSn = (3 << 16) | i16; CIn = CI1[r]; CDn = CD1[r]; }
                            }
                            CS1[r] = CSn; CI1[r] = CIn; CD1[r] = CDn;
                            if (inband && k >= marker) s_cd[par][i - Lk] = CDn;
                        }
                        S1[r] = Sv; I1[r] = Iv; D1[r] = Dv;
                        if (k <= marker) {                                                      // :548-557
                            const uint32_t gapState = gapIsI ? 1u : 2u;
                            const uint32_t nib = (isM ? 0u : gapState) | (Iptr ? 4u : 0u) | (Dptr ? 8u : 0u);
                            tbacc[r] |= nib << (4 * (k & 7));
                            tbdirty[r] = true;
                        }
                        // wave summaries -> LDS reduction slot of this diagonal
                        const float Sin = inband ? Sv : -inf;             // out-of-band lanes take no part in the reductions
                        const unsigned long long vm = __builtin_amdgcn_ballot_w64(Sin > -inf);
                        const bool raise = __builtin_amdgcn_ballot_w64(Sin > msp) != 0ull;
                        int wkey = (int)0x80000000;                      // neutral for max
                        if (raise) wkey = f2key(wave_max_f32(Sin));
                        const int firstRow = vm ? b + (int)__builtin_ctzll(vm) : 0x7fffffff;     // neutral for min
                        const int lastRow = vm ? b + 63 - (int)__builtin_clzll(vm) : -1;          // neutral for max
                        if (lane == 63) {        // the block's single-lane side effects: three reductions and the mailbox for block b+64
                            lds_max_i32(&s_red[rs3][0], wkey);
                            lds_min_i32(&s_red[rs3][1], firstRow);
                            lds_max_i32(&s_red[rs3][2], lastRow);
                            s_exch[par][vw] = make_int4(__float_as_int(Sv), __float_as_int(Iv), CS1[r], CI1[r]);
                        }
                        LS2[r] = LS1; LCS2[r] = LCS1;
                    }
                    if (tbdirty[r] && (((k & 7) == 7) || k == marker)) {
                        tb[(size_t)(k >> 3) * WINDOW + 64 * vw + lane] = tbacc[r];
                        tbacc[r] = 0; tbdirty[r] = false;
                    }
                    uph[r] += 1; if (uph[r] == CAP) uph[r] = 0;
                }
                heartbeat(a, 4, k);
                TWL_STAMP(t_slots);
                __syncthreads();
                TWL_STAMP(t_bar);
                heartbeat(a, 5, k);
                const int rs3_now = rs3;
                rs3 = (rs3 == 2) ? 0 : rs3 + 1;

                const int gkey = __builtin_amdgcn_readfirstlane(s_red[rs3_now][0]);
                const int gfirst = __builtin_amdgcn_readfirstlane(s_red[rs3_now][1]);
                const int glast = __builtin_amdgcn_readfirstlane(s_red[rs3_now][2]);
                const bool anyValid = glast >= 0;
                const int newL = anyValid ? gfirst : Uk + 1;                  // :563-583
                const int newU = anyValid ? glast : Lk - 1;
                mspKey = max(mspKey, gkey);                                   // :501-503

                // Before diagonal marker-1 every CS entry is the initial -1, so conv_S = -1 and nothing can converge (:585-595
                // reduce to prev_conv_s = -1, which it already is): the whole test is skipped with one branch.
                if (__builtin_expect(pb && !converged && k < kEnd - 1, 0)) {   // :585-595
                    int conv_S = -1;
                    bool all3 = false;
                    if (k == marker - 1) conv_S = (newL == newU) ? ((3 << 16) | (newL & 0xFFFF)) : -1;
                    else if (k == marker) conv_S = (newL == newU) ? (newL & 0xFFFF) : -1;
                    else if (k > marker) {
                        if (threadIdx.x == 0) {
                            s_conv[par ^ 1][0] = 0x7fffffff; s_conv[par ^ 1][1] = (int)0x80000000; s_conv[par ^ 1][2] = 0;
                        }
#pragma unroll
                        for (int r = 0; r < RPL; ++r) {
                            const int b = 64 * blk[r];
                            if (b <= newU && b + 63 >= newL) {
                                const int i = b + lane;
                                const bool inr = (i >= newL) && (i <= newU);
                                const unsigned long long rm = __builtin_amdgcn_ballot_w64(inr);
                                if (rm) {
                                    const int fl = (int)__builtin_ctzll(rm);
                                    const int v = __builtin_amdgcn_readlane(CS1[r], fl);
                                    const bool badS = __builtin_amdgcn_ballot_w64(inr && CS1[r] != v) != 0ull;
                                    const bool badID = __builtin_amdgcn_ballot_w64(inr && (CI1[r] != v || CD1[r] != v)) != 0ull;
                                    if (lane == 0) {
                                        lds_min_i32(&s_conv[par][0], v);
                                        lds_max_i32(&s_conv[par][1], v);
                                        if (badS || badID) lds_or_b32(&s_conv[par][2], (badS ? 1 : 0) | (badID ? 2 : 0));
                                    }
                                }
                            }
                        }
                        __syncthreads();
                        const int vmin = __builtin_amdgcn_readfirstlane(s_conv[par][0]);
                        const int vmax = __builtin_amdgcn_readfirstlane(s_conv[par][1]);
                        const int fl = __builtin_amdgcn_readfirstlane(s_conv[par][2]);
                        if (newU >= newL && vmin == vmax && !(fl & 1)) { conv_S = vmin; all3 = !(fl & 2); }
                    }
                    if (all3 && prev_conv_s == conv_S && conv_S != -1) {
                        converged = true; conv_value = prev_conv_s; convKey = mspKey;
                    }
                    prev_conv_s = conv_S;
                }

                {                                                             // :597-604
                    const int v2 = k + 2 - refLen;
                    const int Lprime = v2 > 0 ? v2 : 0;
                    const int nL = newL > Lprime ? newL : Lprime;
                    const int nU = (qLen - 1) < (newU + 1) ? (qLen - 1) : (newU + 1);
                    L2 = L1; U2 = U1; L1 = Lk; U1 = Uk; Lk = nL; Uk = nU;
                }
                msKey = max(mspKey, 0);                                       // :607  max(0, max_score_prime)
                last_k = k;
#ifdef TWL_KERNEL_STAMPS
                {
                    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
                    st_slots += t_slots - t_head; st_bar += t_bar - t_slots; st_post += t_end - t_bar; st_n += 1;
                    st_act += (64 * blk[0] <= Uk + 1 && 64 * blk[0] + 63 >= Lk) ? 1 : 0;
                }
#endif
                if (converged && msKey > convKey) { conv_logic = true; break; }          // :609-612
            }

            TWL_STAMP(t_exit0);
            cells += tile_cells;
            dbg_lastk = last_k; dbg_conv = conv_value; dbg_L = Lk; dbg_U = Uk;
            if (tile_err != 0) { err = tile_err; break; }

            // a tile that ends before the marker leaves its last (partial) group of 8 diagonals unflushed
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                if (tbdirty[r]) { tb[(size_t)(last_k >> 3) * WINDOW + 64 * (r * W + w) + lane] = tbacc[r]; tbdirty[r] = false; }
            }

            // ---- tile exit, :615-682 ----
            // after the loop (L1,U1) is the band of diagonal last_k (rotated once more at its end)
            int conv_q = 0, conv_r = 0, tb_state = 0, start_k = 0;
            bool bad = false;
            if (!conv_logic && last_k >= marker) {                            // :633-635 needs CS[last_k][0]
                const int Llast = L1;
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int b = 64 * blk[r];
                    if (Llast >= b && Llast <= b + 63 && lane == Llast - b) s_misc[1] = CS1[r];
                }
                __syncthreads();
                conv_value = __builtin_amdgcn_readfirstlane(s_misc[1]);
            }
            if (conv_logic || last_k >= marker) {
                conv_q = conv_value & 0xFFFF;
                tb_state = (conv_value >> 16) & 0xFFFF;
                if (tb_state > 3) bad = true;      // boundary sentinel / unset: the reference indexes out of range here
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

            heartbeat(a, 6, tile);
            __syncthreads();   // all traceback-pointer stores of this tile are complete and visible
            heartbeat(a, 7, tile);
            if (w == 0) {
                int n = 0;
                if (lane == 0) {   // Traceback, :134-231, addressed by (diagonal, row) instead of a ragged offset
                    int kk = start_k, ii = conv_q, qi = conv_q, ri = conv_r, st = tb_state % 3;
                    const bool first = (tile == 0);
                    while (kk >= 0) {
                        const uint32_t word = __hip_atomic_load(&tb[(size_t)(kk >> 3) * WINDOW + (ii % WINDOW)], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                        const int v = (int)((word >> (4 * (kk & 7))) & 0xFu);
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
                        s_rev[n++] = (int8_t)dir;
                        if (first && (ri < 0 || qi < 0)) break;
                        if (ii < 0) break;   // defensive: a pointer chain left the tile (never on valid data)
                    }
                    if (first) {
                        while (ri > -1) { s_rev[n++] = 2; ri--; }
                        while (qi > -1) { s_rev[n++] = 1; qi--; }
                    }
                }
                n = __builtin_amdgcn_readfirstlane(n);
                heartbeat(a, 8, n);
                const int skip = (tile > 0) ? 1 : 0;                          // :98-102
                const int cnt = n - skip;
                if (pos + cnt + tailLen > 2 * a.seq_len) { err = 3; }
                else {
                    for (int t = lane; t < cnt; t += 64) out[pos + t] = s_rev[n - 1 - skip - t];
                    for (int t = lane; t < tailLen; t += 64) out[pos + cnt + t] = (int8_t)tailDir;
                    pos += cnt + tailLen;
                }
                if (lane == 0) s_misc[2] = err;
            }
            __syncthreads();
            err = __builtin_amdgcn_readfirstlane(s_misc[2] == 3 ? 3 : err);
#ifdef TWL_KERNEL_STAMPS
            st_exit += __builtin_amdgcn_s_memtime() - t_exit0;
#endif
            if (err != 0) break;
            ++tile;
        }

        heartbeat(a, 9, err);
#ifdef TWL_KERNEL_STAMPS
        if (a.dbg && lane == 0 && pair == 0) {      // per wave of the workgroup that aligned pair 0: cycle sums per segment
            long long *g = reinterpret_cast<long long *>(a.dbg + 16 * (size_t)a.n_pairs_total) + 8 * w;
            g[0] = (long long)st_slots; g[1] = (long long)st_bar; g[2] = (long long)st_post; g[3] = (long long)st_n; g[4] = (long long)st_act;
            g[5] = (long long)(__builtin_amdgcn_s_memtime() - st_t0);
            g[6] = (long long)st_exit; g[7] = (long long)st_setup;
        }
#endif
        __syncthreads();
        if (threadIdx.x == 0) {
            a.err[pair] = (int16_t)err;
            a.aln_len[pair] = (err == 0) ? pos : 0;
            a.cells[pair] = cells;
            heartbeat(a, 10, pair);
            if (a.dbg) {
                int32_t *g = a.dbg + 16 * (size_t)pair;
                g[0] = tile; g[1] = dbg_lastk; g[2] = dbg_conv; g[3] = dbg_L; g[4] = dbg_U; g[5] = ref_idx; g[6] = qry_idx;
                g[7] = pos; g[8] = err; g[9] = steps_left; g[10] = R; g[11] = Q;
            }
        }
        __syncthreads();   // keeps the single-lane block above out of the loop latch (see NOTE on control flow)
    }
}

// ---- precomputed protein column scores (matrix mode 4) ----
// One workgroup = 64 query rows x 64 anti-diagonals of one pair; lane <-> query row, so the stores of one anti-diagonal are
// consecutive floats, the order the DP kernel reads them in.  Arithmetic and summation order are those of the in-kernel sparse
// path (and therefore of TALCO-XDrop.cpp:409-444): letters of the reference column in ascending order, per letter the tail
// m = 16..20 first, then the eight pair sums left to right; then the two gap-letter loops (:432,:433); then the division (:444).
struct ScoreArgs {
    const float *cols;
    const int32_t *len, *num;
    const int32_t *items;          // pair ids
    const int32_t *blk_off;        // [n_items + 1] first workgroup of each item
    int32_t n_items, seq_len;
    float gap_char;
    const uint8_t *gc_zero;        // optional [pair], as in KArgs
    const float *M24;              // P = 22: [21][24] matrix rows padded to 24 floats; P = 6: the 5x5 matrix row-major, padded to 28 floats
    float *sim;
    const long long *sim_off;
    int32_t corridor;              // > 0: only the 64 x 64 tiles within this many rows of the straight line between the corners are scored, the others filled with NaN (round 6)
};

template <int P>
__global__ void __launch_bounds__(256) score_matrix_kernel(ScoreArgs a)
{
    constexpr int CW = P + 2, F4 = CW / 4, MS = P - 1;
    __shared__ float4 s_r[F4 * 128];           // plane-major: reference columns Jmin .. Jmin+126
    __shared__ uint32_t s_mask[128];
    __shared__ float4 s_M4[P == 22 ? 21 * 6 : 8];      // P = 22: rows padded to 24 floats; P = 6: the 25 values, row-major
    int it = 0;
    while (it + 1 < a.n_items && (int)blockIdx.x >= a.blk_off[it + 1]) ++it;
    const int pair = a.items[it];
    const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
    const int nI = (Q + 63) >> 6;
    const int local = (int)blockIdx.x - a.blk_off[it];
    const int I0 = (local % nI) * 64, K0 = (local / nI) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Round 6: the DP band of a pair is 27-37 % of R x Q on 2 kaa profiles and follows the straight line between the corners to within a few dozen rows; tiles further
    // than `corridor` rows from that line are not scored.  They are filled with NaN instead, and a DP kernel whose band reads one hands the pair to the kernel that
    // scores in line (talco_lean_kernel, PRESIM: kErrGuard) -- a prediction like the scouts': results do not depend on it.
    if (a.corridor > 0) {
        const int ic = (int)(((long long)(K0 + 32) * Q) / (R + Q));
        if (I0 > ic + a.corridor || I0 + 63 < ic - a.corridor) {
            const int pitchN = (Q + 63) & ~63;
            float *outN = a.sim + a.sim_off[pair];
            const int I = I0 + lane;
            for (int kk = wave; kk < 64; kk += 4) {
                const int K = K0 + kk;
                if (I < pitchN && K < R + Q) outN[(size_t)K * (size_t)pitchN + (size_t)I] = __int_as_float(0x7fc00000);
            }
            return;
        }
    }
    const int Jmin = K0 - I0 - 63;
    const float4 *colsR = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * CW);
    const float4 *colsQ = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * CW);
    for (int t = threadIdx.x; t < (P == 22 ? 21 * 6 : 7); t += 256) s_M4[t] = reinterpret_cast<const float4 *>(a.M24)[t];
    if (threadIdx.x < 127) {
        const int J = Jmin + (int)threadIdx.x;
        uint32_t mk = 0;
#pragma unroll
        for (int t = 0; t < F4; ++t) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (J >= 0 && J < R) c = colsR[F4 * (size_t)J + t];
            s_r[t * 128 + threadIdx.x] = c;
            if (4 * t + 0 < MS) mk |= (c.x != 0.0f) ? (1u << (4 * t + 0)) : 0u;
            if (4 * t + 1 < MS) mk |= (c.y != 0.0f) ? (1u << (4 * t + 1)) : 0u;
            if (4 * t + 2 < MS) mk |= (c.z != 0.0f) ? (1u << (4 * t + 2)) : 0u;
            if (4 * t + 3 < MS) mk |= (c.w != 0.0f) ? (1u << (4 * t + 3)) : 0u;
        }
        s_mask[threadIdx.x] = mk;
    }
    const int I = I0 + lane;
    float q[CW];
#pragma unroll
    for (int t = 0; t < F4; ++t) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (I < Q) c = colsQ[F4 * (size_t)I + t];
        q[4 * t] = c.x; q[4 * t + 1] = c.y; q[4 * t + 2] = c.z; q[4 * t + 3] = c.w;
    }
    __syncthreads();
    const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];
    const bool denomOne = (denom == 1.0f);
    const float gc = (a.gc_zero && a.gc_zero[pair]) ? 0.0f : a.gap_char;
    const int pitch = (Q + 63) & ~63;
    float *out = a.sim + a.sim_off[pair];
    const float *rf = reinterpret_cast<const float *>(s_r);
    const float *Mf = reinterpret_cast<const float *>(s_M4);
    for (int kk = wave; kk < 64; kk += 4) {
        const int K = K0 + kk;
        const int J = K - I;
        const bool ok = (I < Q) && (J >= 0) && (J < R);
        const int jr = ok ? J - Jmin : 0;              // 0..126 when ok
        float numer = 0.0f;
        if constexpr (P == 22) {
            uint32_t mk = ok ? s_mask[jr] : 0u;
            const uint32_t mk0 = mk;
            while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                if (mk != 0u) {
                    const int l = __builtin_ctz(mk);
                    mk &= mk - 1u;
                    const float rl = rf[(size_t)((l >> 2) * 128 + jr) * 4 + (l & 3)];
                    float Mr[24];
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const float4 c = s_M4[l * 6 + t];
                        Mr[4 * t] = c.x; Mr[4 * t + 1] = c.y; Mr[4 * t + 2] = c.z; Mr[4 * t + 3] = c.w;
                    }
#pragma unroll
                    for (int m = 16; m < 21; ++m) numer += (rl * q[m]) * Mr[m];
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = (q[t] * Mr[t]) * rl + (q[8 + t] * Mr[8 + t]) * rl;
                    numer += ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
                }
            }
            // gap-letter terms: (r[l]*q[gap])*gc over l (:432) and (r[gap]*q[m])*gc over m (:433); terms with a zero factor are +-0
            mk = mk0;
            while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                if (mk != 0u) {
                    const int l = __builtin_ctz(mk);
                    mk &= mk - 1u;
                    numer += (rf[(size_t)((l >> 2) * 128 + jr) * 4 + (l & 3)] * q[P - 1]) * gc;
                }
            }
        } else {
            // nucleotide column score, TALCO-XDrop.cpp:378-395: per l the five products summed left to right, accumulated over l
            float rc[5];
#pragma unroll
            for (int l = 0; l < 5; ++l) rc[l] = rf[(size_t)((l >> 2) * 128 + jr) * 4 + (l & 3)];
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                float t[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) t[m] = (q[m] * Mf[5 * l + m]) * rc[l];
                const float sl = (((t[0] + t[1]) + t[2]) + t[3]) + t[4];
                numer = (l == 0) ? sl : numer + sl;
            }
#pragma unroll
            for (int l = 0; l < 5; ++l) numer += (rc[l] * q[P - 1]) * gc;                       // :394
        }
        if (ok) {
            const float rg = rf[(size_t)(((P - 1) >> 2) * 128 + jr) * 4 + ((P - 1) & 3)];
#pragma unroll
            for (int m = 0; m < MS; ++m) numer += (rg * q[m]) * gc;                             // :395 / :433
            out[(size_t)K * (size_t)pitch + (size_t)I] = denomOne ? numer : numer / denom;
        }
    }
}

// ---- column packing: freq[pair][2][seq_len][P] + gapOpen/gapExtend[pair][2][seq_len] -> cols[pair][2][seq_len][P+2] ----
template <int P>
__global__ void pack_kernel(const float *__restrict__ freq, const float *__restrict__ gop, const float *__restrict__ gex,
                            float *__restrict__ cols, size_t n_cols)
{
    constexpr int CW = P + 2;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_cols; c += (size_t)gridDim.x * blockDim.x) {
        const float *f = freq + c * P;
        float4 *dst = reinterpret_cast<float4 *>(cols + c * CW);
        float v[CW];
#pragma unroll
        for (int t = 0; t < P; ++t) v[t] = f[t];
        v[P] = gop[c]; v[P + 1] = gex[c];
#pragma unroll
        for (int t = 0; t < CW / 4; ++t) dst[t] = make_float4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
    }
}

}  // namespace twl

// twilight_amd/csrc/restore_kernels.hip.h -- alignment_helper::addGappyColumnsBack + pairwiseGlobal on the device
// (/root/reference/src/alignment-helper.cpp:324-375, :243-322), so that the paths of a level never leave HBM.
//
// The reference walks the DP path once, keeping the ORIGINAL column index of either side, and inserts a removed run of columns when
// that index reaches the run's start: 2s for a reference run, 1s for a query run, and when both sides have a run at the same step
// the two runs (their consensus letters) are aligned to each other by a small affine Needleman-Wunsch.  As a data-parallel program:
//   * a removed run of side s lies immediately before kept column j of that side (or after the last one); its length is
//     orig_idx[j] - orig_idx[j-1] - 1, where orig_idx[j] is the original index of kept column j (restore_index_kernel);
//   * "the index reaches the run's start" at path boundary a (between elements a-1 and a) iff a == 0 or element a-1 consumed a column
//     of that side, and then j = the number of columns of that side the path has consumed before a (restore_runs_kernel: two scans);
//   * boundaries with a run on both sides are aligned by ONE thread each (restore_align_kernel; the runs are a few columns long,
//     thousands of them per pair at the top of a tree), same operations in the same order as the host mirror's pairwiseGlobal
//     (twilight_amd/csrc/host/helpers.cpp); a pair with a two-sided boundary too large for the per-thread scratch is flagged and
//     restored by the host instead;
//   * the output position of boundary a's segment is a + (lengths of the segments before it): one more scan (restore_write_kernel).
// Checked against the host mirror on every level of the end-to-end tests (tests/test_gpu_level.py, tests/test_gpu_msa.py).
#pragma once
#include "level_kernels.hip.h"

namespace twl {

constexpr int kNwCells = 4096;      // (m + 1) * (n + 1) of a two-sided boundary one thread aligns
constexpr int kNwRow = 128;         // n + 1

struct RestoreArgs {
    const int8_t *aln;        // [n_pairs][aln_stride] DP paths of the level (codes 0 / 1 / 2)
    const int32_t *aln_len;   // [n_pairs]
    int32_t aln_stride;
    const uint8_t *colinfo;   // [2 * n_pairs][stride] consensus letter index | 0x80 when the column was removed
    int32_t stride;
    const SideDesc *sides;    // .len = columns before removal
    const int32_t *len_red;   // [2 * n_pairs] columns after removal
    const int32_t *sel;       // [n_sel] the pairs to restore
    int32_t n_sel;
    // work arrays, one slot per SELECTED pair (slot = index into sel)
    int32_t *orig_idx;        // [2 * n_sel][stride + 1]
    int32_t *run;             // [n_sel][4][bstride] per boundary: run length ref, run length query, run start ref, run start query
    int32_t *seg;             // [n_sel][bstride] segment length per boundary
    int32_t *aoff;            // [n_sel][bstride] arena offset of a two-sided boundary
    int32_t *both_list;       // [n_sel][bstride] the two-sided boundaries of a pair, ascending
    int32_t *n_both;          // [n_sel] how many
    int32_t *wtot;            // [n_sel][n_wchunks] bytes of the final path per chunk of boundaries (restore_count_kernel -> restore_write_kernel)
    int32_t n_wchunks;
    int8_t *arena;            // [n_sel][out_stride] the aligned two-sided segments, reversed
    int32_t bstride;
    int8_t *out;              // [n_pairs][out_stride] final paths
    int32_t out_stride;
    int32_t *out_len;         // [n_pairs] final length; -1 = the host has to restore this pair
    int8_t *tbs;              // [n_sel * nb * 256][kNwCells] per-thread traceback scratch of restore_align_kernel
    float *rows;              // [n_sel * nb * 256][6 * kNwRow] per-thread rolling rows
    float M[441];             // scoringMatrix[a][b], row-major ms x ms
    int32_t ms;
    float gap_open, gap_extend;
};

// The three scan kernels below run one workgroup of kRsThreads threads per pair (or side), kRsItems consecutive items per thread and
// one workgroup scan per kRsThreads * kRsItems items: at the top of a tree a level is ONE pair with a path of 10^5 elements, and the
// time of these kernels is the number of scan rounds.
constexpr int kRsThreads = 1024, kRsItems = 16, kRsWaves = kRsThreads / 64;
// Item loads are UNCONDITIONAL with clamped indices and the values are masked afterwards: a load under a lane condition compiles to a
// branch with its own wait, and the sixteen items of a thread then cost sixteen memory round trips instead of one.
constexpr int kRsItemsRuns = 8;     // restore_runs_kernel keeps four values per item: 16 items would not fit the 128 registers of a 1024-thread workgroup

// grid: 2 * n_sel workgroups: original index of every kept column of one side (+ the side's original length as a sentinel)
__global__ void __launch_bounds__(kRsThreads) restore_index_kernel(RestoreArgs a)
{
    __shared__ int s_wave[kRsWaves];
    const int side = 2 * a.sel[blockIdx.x >> 1] + (blockIdx.x & 1);
    const int len = a.sides[side].len;
    const uint8_t *ci = a.colinfo + (size_t)side * a.stride;
    int32_t *oi = a.orig_idx + (size_t)blockIdx.x * (size_t)(a.stride + 1);
    int base = 0;
    for (int c0 = 0; c0 < len; c0 += kRsThreads * kRsItems) {
        const int t0 = c0 + (int)threadIdx.x * kRsItems;
        bool keep[kRsItems];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kRsItems; ++k) { const uint8_t f = ci[min(t0 + k, len - 1)]; keep[k] = (t0 + k < len) & !(f & 0x80); cnt += keep[k] ? 1 : 0; }
        int total;
        int dst = base + block_scan_int<kRsWaves>(cnt, &total, s_wave);
#pragma unroll
        for (int k = 0; k < kRsItems; ++k) if (keep[k]) oi[dst++] = t0 + k;
        base += total;
    }
    if (threadIdx.x == 0) oi[base] = len;      // (base == len_red[side])
}

// grid: n_sel workgroups: the runs at every boundary of the DP path, where the two-sided ones go in the arena, and the list of them
__global__ void __launch_bounds__(kRsThreads) restore_runs_kernel(RestoreArgs a)
{
    __shared__ int s_wave[kRsWaves];
    __shared__ int s_big;
    const int pair = a.sel[blockIdx.x];
    const int n = a.aln_len[pair];
    const int8_t *path = a.aln + (size_t)pair * a.aln_stride;
    const int32_t *oiR = a.orig_idx + (size_t)(2 * blockIdx.x) * (size_t)(a.stride + 1), *oiQ = oiR + (a.stride + 1);
    int32_t *runR = a.run + (size_t)blockIdx.x * 4 * a.bstride, *runQ = runR + a.bstride, *stR = runQ + a.bstride, *stQ = stR + a.bstride;
    int32_t *seg = a.seg + (size_t)blockIdx.x * a.bstride, *aoff = a.aoff + (size_t)blockIdx.x * a.bstride;
    int32_t *both_list = a.both_list + (size_t)blockIdx.x * a.bstride;
    if (threadIdx.x == 0) s_big = 0;
    int baseR = 0, baseQ = 0, baseA = 0, baseB = 0;
    bool tooBig = false;
    for (int c0 = 0; c0 <= n; c0 += kRsThreads * kRsItemsRuns) {
        const int b0 = c0 + (int)threadIdx.x * kRsItemsRuns;       // boundary b: between elements b - 1 and b
        bool fR[kRsItemsRuns], fQ[kRsItemsRuns];
        int nR = 0, nQ = 0;
#pragma unroll
        for (int k = 0; k < kRsItemsRuns; ++k) {
            const int b = b0 + k;
            const int pv = path[max(min(b, n) - 1, 0)];
            const int prev = (b >= 1 && b <= n) ? pv : 3;
            fR[k] = (prev == 0) | (prev == 2); fQ[k] = (prev == 0) | (prev == 1);
            nR += fR[k] ? 1 : 0; nQ += fQ[k] ? 1 : 0;
        }
        int totR, totQ, totA, totB;
        int cR = baseR + block_scan_int<kRsWaves>(nR, &totR, s_wave);      // reference columns consumed before boundary b0 (element b0 - 1 not counted yet)
        int cQ = baseQ + block_scan_int<kRsWaves>(nQ, &totQ, s_wave);
        int rR[kRsItemsRuns], rQ[kRsItemsRuns], sR[kRsItemsRuns], sQ[kRsItemsRuns];
        int sumA = 0, nB = 0;
#pragma unroll
        for (int k = 0; k < kRsItemsRuns; ++k) {
            const int b = b0 + k;
            cR += fR[k] ? 1 : 0; cQ += fQ[k] ? 1 : 0;                     // ... consumed before boundary b
            // (cR <= the side's kept columns, whose sentinel entry orig_idx[kept] exists: the loads are in range for every lane)
            const int iR = min(cR, a.stride), iQ = min(cQ, a.stride);
            const int pR = oiR[max(iR - 1, 0)], qR = oiR[iR], pQ = oiQ[max(iQ - 1, 0)], qQ = oiQ[iQ];
            const bool useR = (b <= n) & ((b == 0) | fR[k]), useQ = (b <= n) & ((b == 0) | fQ[k]);
            sR[k] = useR ? (cR ? pR + 1 : 0) : 0; rR[k] = useR ? qR - sR[k] : 0;
            sQ[k] = useQ ? (cQ ? pQ + 1 : 0) : 0; rQ[k] = useQ ? qQ - sQ[k] : 0;
            const bool both = rR[k] > 0 && rQ[k] > 0;
            if (both && ((long long)(rR[k] + 1) * (rQ[k] + 1) > kNwCells || rQ[k] + 1 > kNwRow)) tooBig = true;
            sumA += both ? rR[k] + rQ[k] : 0;
            nB += both ? 1 : 0;
        }
        int off = baseA + block_scan_int<kRsWaves>(sumA, &totA, s_wave);
        int lst = baseB + block_scan_int<kRsWaves>(nB, &totB, s_wave);
#pragma unroll
        for (int k = 0; k < kRsItemsRuns; ++k) {
            const int b = b0 + k;
            if (b > n) continue;
            const bool both = rR[k] > 0 && rQ[k] > 0;
            runR[b] = rR[k]; runQ[b] = rQ[k]; stR[b] = sR[k]; stQ[b] = sQ[k];
            seg[b] = both ? -1 : rR[k] + rQ[k];
            aoff[b] = off;
            if (both) { both_list[lst++] = b; off += rR[k] + rQ[k]; }
        }
        baseR += totR; baseQ += totQ; baseA += totA; baseB += totB;
    }
    __syncthreads();
    if (tooBig) s_big = 1;
    __syncthreads();
    if (threadIdx.x == 0) { a.out_len[pair] = s_big ? -1 : 0; a.n_both[blockIdx.x] = baseB; }
}

// One small affine Needleman-Wunsch (pairwiseGlobal, helpers.cpp / alignment-helper.cpp:243-322) by one thread: rolling rows of M / X / Y
// and the traceback matrix live in the storage the accessors address (per-thread LDS for runs of up to kNwSmall columns, global scratch
// beyond); the operations and their order are the same either way.  Writes the reversed path to dst, returns its length.
constexpr int kNwSmall = 31;        // runs of up to this many columns on both sides: rows and traceback in LDS (one wave per workgroup)
template <class L1, class L2, class Rows, class Tb>
__device__ __forceinline__ int nw_small(L1 s1, int m, L2 s2, int nn, const float *Mx, int ms, float go, float ge, Rows R, Tb T, int8_t *dst)
{
    const int W = nn + 1;
    int Mp = 0, Xp = 1, Yp = 2, Mc = 3, Xc = 4, Yc = 5;
    R(Mp, 0) = 0.0f; R(Xp, 0) = 0.0f; R(Yp, 0) = 0.0f; T(0) = 0;
    for (int j = 1; j <= nn; ++j) { R(Mp, j) = 0.0f; R(Yp, j) = 0.0f; R(Xp, j) = -1e9f; T(j) = 1; }
    for (int i = 1; i <= m; ++i) {
        const float *row = Mx + (size_t)(s1(i - 1) & 0x7f) * ms;
        R(Mc, 0) = 0.0f; R(Xc, 0) = 0.0f; R(Yc, 0) = -1e9f; T(i * W) = 2;
        // (the cell's left and upper-left neighbours travel in registers: three reads of the row above per cell)
        float dM = R(Mp, 0), dX = R(Xp, 0), dY = R(Yp, 0), lM = 0.0f, lY = -1e9f;
        for (int j = 1; j <= nn; ++j) {
            const float base = row[s2(j - 1) & 0x7f];
            const float uM = R(Mp, j), uX = R(Xp, j), uY = R(Yp, j);
            const float mv = base + fmaxf(fmaxf(dM, dX), dY);
            const float xv = fmaxf(uM + go, uX + ge);
            const float yv = fmaxf(lM + go, lY + ge);
            const float best = fmaxf(fmaxf(mv, xv), yv);
            R(Mc, j) = mv; R(Xc, j) = xv; R(Yc, j) = yv;
            T(i * W + j) = (best == mv) ? 0 : ((best == yv) ? 1 : 2);
            dM = uM; dX = uX; dY = uY; lM = mv; lY = yv;
        }
        int t;
        t = Mp; Mp = Mc; Mc = t; t = Xp; Xp = Xc; Xc = t; t = Yp; Yp = Yc; Yc = t;
    }
    int len = 0;
    for (int i = m, j = nn; i > 0 || j > 0;) {
        const int8_t d = T(i * W + j);
        dst[len++] = d;
        if (d == 0) { --i; --j; }
        else if (d == 1) --j;
        else --i;
    }
    return len;
}

// grid: (n_sel, nb) workgroups of ONE wave: every two-sided boundary aligned by one thread; the boundaries of a pair (restore_runs_kernel
// listed them) are dealt to the nb * 64 threads of its workgroups (thousands of small alignments per pair at the top of a tree, the largest
// of them a few hundred cells: what the launch takes is its longest alignment, so the rows and the traceback of all but the rare large
// ones live in LDS, 1.8 KB per thread)
constexpr int kNwThreads = 64;
__global__ void __launch_bounds__(kNwThreads) restore_align_kernel(RestoreArgs a)
{
    __shared__ float s_rows[6 * (kNwSmall + 1) * kNwThreads];                      // [row][j][thread]
    __shared__ int8_t s_tb[(kNwSmall + 1) * (kNwSmall + 1) * kNwThreads];          // [cell][thread]
    __shared__ uint8_t s_l1[kNwSmall * kNwThreads], s_l2[kNwSmall * kNwThreads];   // the two runs' consensus letters
    __shared__ float s_M[441];
    for (int t = threadIdx.x; t < a.ms * a.ms; t += kNwThreads) s_M[t] = a.M[t];
    __syncthreads();
    const int pair = a.sel[blockIdx.x];
    if (a.out_len[pair] < 0) return;
    const int32_t *runR = a.run + (size_t)blockIdx.x * 4 * a.bstride, *runQ = runR + a.bstride, *stR = runQ + a.bstride, *stQ = stR + a.bstride;
    int32_t *seg = a.seg + (size_t)blockIdx.x * a.bstride;
    const int32_t *aoff = a.aoff + (size_t)blockIdx.x * a.bstride;
    const uint8_t *cR = a.colinfo + (size_t)(2 * pair) * a.stride, *cQ = cR + a.stride;
    int8_t *arena = a.arena + (size_t)blockIdx.x * a.out_stride;
    const size_t thr = ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * kNwThreads + threadIdx.x;
    int8_t *tb = a.tbs + thr * kNwCells;
    float *rw = a.rows + thr * (6 * kNwRow);
    const int32_t *both_list = a.both_list + (size_t)blockIdx.x * a.bstride;
    const int nBoth = a.n_both[blockIdx.x];
    const int tid = threadIdx.x;
    for (int t = blockIdx.y * kNwThreads + threadIdx.x; t < nBoth; t += kNwThreads * gridDim.y) {
        const int b = both_list[t];
        const int m = runR[b], nn = runQ[b];
        const uint8_t *s1 = cR + stR[b], *s2 = cQ + stQ[b];
        int8_t *dst = arena + aoff[b];       // reversed: the write kernel turns it round
        if (m <= kNwSmall && nn <= kNwSmall) {      // letters, scores, rows and traceback in LDS: no global load inside the cell loop
            for (int i = 0; i < m; ++i) s_l1[i * kNwThreads + tid] = s1[i];
            for (int j = 0; j < nn; ++j) s_l2[j * kNwThreads + tid] = s2[j];
            seg[b] = nw_small([&](int i) -> uint8_t { return s_l1[i * kNwThreads + tid]; }, m, [&](int j) -> uint8_t { return s_l2[j * kNwThreads + tid]; }, nn,
                              s_M, a.ms, a.gap_open, a.gap_extend,
                              [&](int r, int j) -> float & { return s_rows[(r * (kNwSmall + 1) + j) * kNwThreads + tid]; },
                              [&](int c) -> int8_t & { return s_tb[c * kNwThreads + tid]; }, dst);
        } else
            seg[b] = nw_small([&](int i) -> uint8_t { return s1[i]; }, m, [&](int j) -> uint8_t { return s2[j]; }, nn, a.M, a.ms, a.gap_open, a.gap_extend,
                              [&](int r, int j) -> float & { return rw[r * kNwRow + j]; },
                              [&](int c) -> int8_t & { return tb[c]; }, dst);
    }
}

// The write pass gives every boundary its own thread (kWrItems = 1): consecutive lanes then touch consecutive words of the per-boundary
// tables (a thread that owns sixteen consecutive boundaries makes every wave-level load touch 64 cache lines), and a path of 10^4
// boundaries spreads over a dozen CUs instead of one.
constexpr int kWrItems = 1;
// grid: (n_sel, n_wchunks) workgroups: bytes of the final path that the boundaries of one chunk (kRsThreads * kWrItems of them) produce
__global__ void __launch_bounds__(kRsThreads) restore_count_kernel(RestoreArgs a)
{
    __shared__ int s_wave[kRsWaves];
    const int pair = a.sel[blockIdx.x];
    if (a.out_len[pair] < 0) return;
    const int n = a.aln_len[pair];
    const int32_t *seg = a.seg + (size_t)blockIdx.x * a.bstride;
    const int b0 = ((int)blockIdx.y * kRsThreads + (int)threadIdx.x) * kWrItems;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kWrItems; ++k) {
        const int b = b0 + k;
        const int sg = seg[min(b, n)];
        sum += (b <= n) ? sg + (b < n ? 1 : 0) : 0;
    }
    int total;
    (void)block_scan_int<kRsWaves>(sum, &total, s_wave);
    if (threadIdx.x == 0) a.wtot[(size_t)blockIdx.x * a.n_wchunks + blockIdx.y] = total;
}

// grid: (n_sel, n_wchunks) workgroups: the final path = for every boundary its segment, then the path element behind it; a chunk's first
// output position is the sum of the chunks before it (restore_count_kernel)
__global__ void __launch_bounds__(kRsThreads) restore_write_kernel(RestoreArgs a)
{
    __shared__ int s_wave[kRsWaves];
    const int pair = a.sel[blockIdx.x];
    if (a.out_len[pair] < 0) return;
    const int n = a.aln_len[pair];
    const int c0 = (int)blockIdx.y * kRsThreads * kWrItems;
    if (c0 > n) return;
    const int8_t *path = a.aln + (size_t)pair * a.aln_stride;
    const int32_t *runR = a.run + (size_t)blockIdx.x * 4 * a.bstride, *runQ = runR + a.bstride;
    const int32_t *seg = a.seg + (size_t)blockIdx.x * a.bstride, *aoff = a.aoff + (size_t)blockIdx.x * a.bstride;
    const int8_t *arena = a.arena + (size_t)blockIdx.x * a.out_stride;
    int8_t *out = a.out + (size_t)pair * a.out_stride;
    int base = 0;
    for (int c = 0; c < (int)blockIdx.y; ++c) base += a.wtot[(size_t)blockIdx.x * a.n_wchunks + c];
    const int b0 = c0 + (int)threadIdx.x * kWrItems;
    int sl[kWrItems];
    // everything the store loop needs is read first: a load behind a store waits for that store to complete (one counter orders them),
    // which made the sixteen boundaries of a thread cost sixteen round trips
    int8_t pe[kWrItems];           // the path element behind the boundary
    int src[kWrItems];             // >= 0: arena offset of a two-sided segment (reversed copy); -1 / -2: a run of code 1 / 2
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kWrItems; ++k) {
        const int b = b0 + k;
        const int bb = min(b, n);
        const int sg = seg[bb], rr = runR[bb], rq = runQ[bb], ao = aoff[bb];
        pe[k] = path[min(b, max(n - 1, 0))];
        sl[k] = (b <= n) ? sg : 0;
        src[k] = ((rr > 0) & (rq > 0)) ? ao : (rr > 0 ? -2 : -1);
        sum += (b <= n) ? sl[k] + (b < n ? 1 : 0) : 0;
    }
    int total;
    int pos = base + block_scan_int<kRsWaves>(sum, &total, s_wave);
    // Segments of up to kSegDirect bytes are written by the boundary's own thread; a removed run can be thousands of columns long (one
    // thread would write it byte by byte while the launch waits): those are queued and filled by the whole workgroup.
    constexpr int kSegDirect = 32, kQueue = 2048;
    __shared__ int q_n, q_pos[kQueue], q_len[kQueue], q_src[kQueue];       // q_src >= 0: arena offset (reversed copy); -1 / -2: fill with code 1 / 2
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWrItems; ++k) {
        const int b = b0 + k;
        if (b > n) continue;
        const int w = sl[k] + (b < n ? 1 : 0);
        if (pos + w <= a.out_stride) {
            if (sl[k] > 0) {
                const bool both = src[k] >= 0;
                int slot = -1;
                if (sl[k] > kSegDirect) { slot = atomicAdd(&q_n, 1); if (slot >= kQueue) slot = -1; }
                if (slot >= 0) { q_pos[slot] = pos; q_len[slot] = sl[k]; q_src[slot] = src[k]; }
                else if (both) { const int8_t *from = arena + src[k]; for (int t = 0; t < sl[k]; ++t) out[pos + t] = from[sl[k] - 1 - t]; }
                else { const int8_t code = (src[k] == -2) ? 2 : 1; for (int t = 0; t < sl[k]; ++t) out[pos + t] = code; }
            }
            if (b < n) out[pos + sl[k]] = pe[k];
        }
        pos += w;
    }
    __syncthreads();
    const int nq = min(q_n, kQueue);
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x >> 6; e < nq; e += kRsWaves) {      // a wave per queued segment
        const int p0 = q_pos[e], len = q_len[e], from = q_src[e];
        if (from >= 0) for (int t = lane; t < len; t += 64) out[p0 + t] = arena[from + len - 1 - t];
        else { const int8_t code = (from == -2) ? 2 : 1; for (int t = lane; t < len; t += 64) out[p0 + t] = code; }
    }
    // (the chunk that holds the last boundary knows the length of the whole path)
    if (threadIdx.x == 0 && n < c0 + kRsThreads * kWrItems) a.out_len[pair] = (base + total <= a.out_stride) ? base + total : -1;
}

// ---- paths between the level's buffers and one contiguous device block (the exchange between processes) ----
// grid: n rows, 256 threads: row t = len[t] bytes at base + row_off[t]  <->  blk + blk_off[t]
__global__ void __launch_bounds__(256) rows_to_block_kernel(const int8_t *base, const int64_t *row_off, const int32_t *len, int8_t *blk, const int64_t *blk_off)
{
    const int t = blockIdx.x;
    const int8_t *src = base + row_off[t];
    int8_t *dst = blk + blk_off[t];
    for (int i = threadIdx.x; i < len[t]; i += 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) block_to_rows_kernel(int8_t *base, const int64_t *row_off, const int32_t *len, const int8_t *blk, const int64_t *blk_off)
{
    const int t = blockIdx.x;
    int8_t *dst = base + row_off[t];
    const int8_t *src = blk + blk_off[t];
    for (int i = threadIdx.x; i < len[t]; i += 256) dst[i] = src[i];
}

}  // namespace twl

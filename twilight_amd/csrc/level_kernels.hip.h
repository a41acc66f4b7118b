// twilight_amd/csrc/level_kernels.hip.h -- device-side pre/post-processing of one guide-tree level.
//
// The rows of every sequence stay in HBM for the whole run (two planes, ping-pong per sequence like
// SequenceDB::SequenceInfo::alnStorage[2], /root/reference/src/msa.hpp:126); these kernels replace the host work
// the reference does around the DP for every pair of a level:
//   profile_kernel   alignment_helper::calculateProfile   /root/reference/src/alignment-helper.cpp:8-72
//                    + getConsensus (:221-241) + the gappy-column test of removeGappyColumns (:84,105)
//   compact_kernel   removeGappyColumns compaction (:74-166) fused with calculatePSGP (:168-219); writes the packed
//                    [P freq | gapOpen | gapExtend] columns the DP kernel reads (talco_kernel.hip.h), so the level's
//                    profiles never cross PCIe
//   path_scan_kernel / apply_path_kernel   alignment_helper::updateAlignment row rewriting (:389-400,436-447)
//   merge_cache_kernel                      alignment_helper::updateFrequency (:506-539)
// All are HBM-bound byte/float streams: one thread per column, members visited in the reference's order so the
// fp32 sums round identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace twl {

// One side (ref or query) of one pair of the level.
struct SideDesc {
    int32_t n_members;       // rows of this side
    int32_t member_off;      // offset into the member tables
    int32_t len;             // aligned length of those rows before gappy-column removal
    int32_t num;             // alnNum
    float weight;            // alnWeight (groupWeight)
    int32_t cache_slot;      // >= 0: index into the cache pointer table; the cached profile is used instead of the rows
    int32_t store_slot;      // >= 0: store the un-normalised profile there (alignment-helper.cpp:35-40)
    int32_t pad;
};

struct LevelArgs {
    const SideDesc *sides;            // [2 * n_pairs]
    const int32_t *member_seq;        // sequence id per member
    const float *member_w;            // seq.weight / groupWeight * num  (alignment-helper.cpp:27), computed by the caller in fp32
    const uint8_t *member_plane;      // current plane per member
    const char *rows0, *rows1;        // planes [n_seqs][cap]
    int64_t cap;
    float *const *cache;              // cache pointer table
    const uint8_t *lut;               // char -> letter index (letterIdx(type, toupper(c)), scoring-matrix.cpp:26-79)
    float *raw;                       // [2*n_pairs][stride][P] un-compacted profiles
    uint8_t *colinfo;                 // [2*n_pairs][stride]  consensus letter index | 0x80 when the column is gappy
    float *cols;                      // [2*n_pairs][stride][P+2] packed DP columns
    int32_t *len_out;                 // [2*n_pairs] lengths after removal
    int32_t *chunk_cnt;               // [2*n_pairs][n_chunks] columns kept in every 1024-column chunk (profile_kernel -> compact_kernel)
    int32_t n_chunks;
    int32_t stride;
    float gappy_thr;                  // option->gappyVertical
    int32_t remove;                   // 0 when gappy_thr == 1 (removeGappyColumns returns early, :77)
    float gap_open, gap_extend, scale, min_gap_open, min_gap_extend;   // calculatePSGP constants (:171-176)
};

// grid: (2 * n_pairs, ceil(stride / 1024)), 256 threads; thread = FOUR consecutive columns of one side: one 32-bit load per member row
// (rows start at multiples of the planes' pitch, a multiple of 256; bytes behind a row's length are '-' padding and are not stored),
// so the per-member table reads are shared by four columns.  Members are visited in the reference's order: the fp32 sums round identically.
template <int P>
__global__ void __launch_bounds__(256) profile_kernel(LevelArgs a)
{
    __shared__ uint8_t s_lut[256];
    __shared__ int s_kept[4];
    s_lut[threadIdx.x] = a.lut[threadIdx.x];
    __syncthreads();
    const int side = blockIdx.x;
    const SideDesc sd = a.sides[side];
    const int t0 = 4 * (blockIdx.y * 256 + threadIdx.x);
    const int nc = max(0, min(4, sd.len - t0));      // (threads behind the side's end carry nothing, but stay for the count below)
    int kept = 0;
    if (nc > 0) {
    float acc[4][P];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < P; ++v) acc[k][v] = 0.0f;
    const float fnum = (float)sd.num;
    if (sd.cache_slot >= 0) {                                            // :16-21  msaFreq / weight * num
        const float *c = a.cache[sd.cache_slot] + (size_t)t0 * P;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nc) {
#pragma unroll
                for (int v = 0; v < P; ++v) acc[k][v] = c[k * P + v] / sd.weight * fnum;
            }
    } else {                                                             // :23-34  member by member
        // U members per round: their row words are requested together (independent loads in flight), then added in member order
        constexpr int U = (P == 6) ? 16 : 2;      // (a side of 900 rows is 900 dependent round trips otherwise: the adds stay in member order, only the loads overlap)
        auto add = [&](uint32_t four, float w) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int li = s_lut[(four >> (8 * k)) & 0xFFu];
#pragma unroll
                for (int v = 0; v < P; ++v) acc[k][v] += (li == v) ? w : 0.0f;    // x + 0 is exact (accumulators are never -0)
            }
        };
        int m = 0;
        for (; m + U <= sd.n_members; m += U) {
            uint32_t four[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int mi = sd.member_off + m + u;
                const char *row = (a.member_plane[mi] ? a.rows1 : a.rows0) + (size_t)a.member_seq[mi] * a.cap;
                four[u] = *reinterpret_cast<const uint32_t *>(row + t0);
                w[u] = a.member_w[mi];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) add(four[u], w[u]);
        }
        for (; m < sd.n_members; ++m) {
            const int mi = sd.member_off + m;
            const char *row = (a.member_plane[mi] ? a.rows1 : a.rows0) + (size_t)a.member_seq[mi] * a.cap;
            add(*reinterpret_cast<const uint32_t *>(row + t0), a.member_w[mi]);
        }
        if (sd.store_slot >= 0) {                                        // :35-40  cache = profile / num * weight
            float *c = a.cache[sd.store_slot] + (size_t)t0 * P;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nc) {
#pragma unroll
                    for (int v = 0; v < P; ++v) c[k * P + v] = acc[k][v] / fnum * sd.weight;
                }
        }
    }
    float *dst = a.raw + ((size_t)side * a.stride + t0) * P;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= nc) continue;
#pragma unroll
        for (int v = 0; v < P; ++v) dst[k * P + v] = acc[k][v];
        {   // the packed DP column at its UNCOMPACTED position: where no column of the side is removed before it (every side of the wide
            // levels at the bottom of a tree) that is where it belongs, and compact_kernel has nothing to move (same arithmetic as there)
            constexpr int CW = P + 2;
            float pc[CW];
#pragma unroll
            for (int v = 0; v < P; ++v) pc[v] = acc[k][v];
            const float g = acc[k][P - 1];
            if (g > 0) {                                                 // calculatePSGP :186-190 (double arithmetic, narrowed once)
                const double frac = ((double)(fnum - g) * 1.0) / (double)sd.num;
                pc[P] = fminf(a.min_gap_open, (float)((double)(a.gap_open * a.scale) * frac));
                pc[P + 1] = fminf(a.min_gap_extend, (float)((double)a.gap_extend * frac));
            } else {
                pc[P] = a.gap_open;
                pc[P + 1] = a.gap_extend;
            }
            float4 *out = reinterpret_cast<float4 *>(a.cols + ((size_t)side * a.stride + t0 + k) * CW);
#pragma unroll
            for (int j = 0; j < CW / 4; ++j) out[j] = make_float4(pc[4 * j], pc[4 * j + 1], pc[4 * j + 2], pc[4 * j + 3]);
        }
        int best = P - 2;                                                // getConsensus: first strict maximum, all-zero -> N / X
        float bestCount = 0.0f;
#pragma unroll
        for (int v = 0; v < P - 2; ++v)
            if (acc[k][v] > bestCount) { bestCount = acc[k][v]; best = v; }
        const bool gappy = (acc[k][P - 1] / fnum > a.gappy_thr);         // :84,105
        a.colinfo[(size_t)side * a.stride + t0 + k] = (uint8_t)(best | (gappy ? 0x80 : 0));
        kept += (a.remove && gappy) ? 0 : 1;
    }
    }
    // columns this 1024-column chunk keeps: compact_kernel places every chunk without walking the side serially
    {
        int x = kept;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) x += __shfl_xor(x, d, 64);
        if ((threadIdx.x & 63) == 0) s_kept[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) a.chunk_cnt[(size_t)side * a.n_chunks + blockIdx.y] = s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
    }
}

// Workgroup-wide exclusive scan of one flag per thread (256 threads = 4 waves); returns the thread's offset, *total = sum.
__device__ inline int block_scan_256(bool flag, int *total, int *s_wave /*[4]*/)
{
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();                               // s_wave may still be read from the previous round
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_wave[w]; if (w < wave) base += c; sum += c; }
    *total = sum;
    return base + before;
}

// exclusive prefix sum of one int per thread over a 256-thread workgroup; *total = the sum
__device__ __forceinline__ int block_scan_int_256(int v, int *total, int *s_wave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int c = s_wave[w]; if (w < wave) base += c; sum += c; }
    *total = sum;
    return base + x - v;
}

// exclusive prefix sum of one int per thread over a workgroup of NW waves; *total = the sum
template <int NW>
__device__ __forceinline__ int block_scan_int(int v, int *total, int *s_wave /*[NW]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    __syncthreads();                               // s_wave may still be read from the previous round
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    int base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const int c = s_wave[w]; if (w < wave) base += c; sum += c; }
    *total = sum;
    return base + x - v;
}

// grid: (2 * n_pairs, n_chunks), 256 threads: one 1024-column chunk of one side, four columns per thread; the chunk's first output
// index is the sum of the kept-column counts of the chunks before it (profile_kernel left them).
template <int P>
__global__ void __launch_bounds__(256) compact_kernel(LevelArgs a)
{
    constexpr int CW = P + 2;
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int side = blockIdx.x;
    const SideDesc sd = a.sides[side];
    const int c0 = 1024 * blockIdx.y;
    const int lastChunk = (sd.len > 0) ? (sd.len - 1) / 1024 : 0;
    if ((int)blockIdx.y > lastChunk) return;
    const float fnum = (float)sd.num;
    const double dnum = (double)sd.num;
    // output index of this chunk's first kept column
    int part = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.y; c += 256) part += a.chunk_cnt[(size_t)side * a.n_chunks + c];
    int tot;
    (void)block_scan_int_256(part, &tot, s_wave);
    if (threadIdx.x == 0) s_base = tot;
    __syncthreads();
    const int base = s_base;
    const int t0 = c0 + 4 * threadIdx.x;
    bool keep[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + k;
        keep[k] = t < sd.len && !(a.remove && (a.colinfo[(size_t)side * a.stride + t] & 0x80));
        cnt += keep[k] ? 1 : 0;
    }
    int total;
    int dst = base + block_scan_int_256(cnt, &total, s_wave);
    if ((int)blockIdx.y == lastChunk && threadIdx.x == 0) a.len_out[side] = base + total;
    // nothing removed up to the end of this chunk: profile_kernel has put its columns where they belong already
    if (base == c0 && total == min(1024, sd.len - c0)) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!keep[k]) continue;
        const float *src = a.raw + ((size_t)side * a.stride + t0 + k) * P;
        float v[CW];
#pragma unroll
        for (int j = 0; j < P; ++j) v[j] = src[j];
        const float g = v[P - 1];
        if (g > 0) {                                                 // calculatePSGP :186-190 (double arithmetic, narrowed once)
            const double frac = ((double)(fnum - g) * 1.0) / dnum;
            v[P] = fminf(a.min_gap_open, (float)((double)(a.gap_open * a.scale) * frac));
            v[P + 1] = fminf(a.min_gap_extend, (float)((double)a.gap_extend * frac));
        } else {
            v[P] = a.gap_open;
            v[P + 1] = a.gap_extend;
        }
        float4 *out = reinterpret_cast<float4 *>(a.cols + ((size_t)side * a.stride + dst) * CW);
#pragma unroll
        for (int j = 0; j < CW / 4; ++j) out[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        ++dst;
    }
}

// ---- write-back ----
struct CommitArgs {
    const int8_t *paths;          // [n_pairs][path_stride]  final paths (gappy columns restored), codes 0/1/2
    const int32_t *path_len;      // [n_pairs]  0 = pair not committed
    int32_t path_stride;
    int32_t *chunk_base;          // [n_pairs][n_chunks][2]  rows consumed before each 256-code chunk (ref, query)
    int32_t n_chunks;
    const SideDesc *sides;
    const int32_t *member_seq;
    const uint8_t *member_plane;
    char *rows0, *rows1;
    int64_t cap;
    const int32_t *work;          // [n_work][3]  side, first member, member count
    float *const *cache;
    const int32_t *merge;         // [n_merge][4]  pair, ref slot, query slot, destination slot
    const float *merge_w;         // [n_merge][2]  refWeight, qryWeight
    // pairs whose DP path IS the final path (no gappy column was removed) are read where the DP kernel left them
    const uint8_t *from_dp;       // optional [n_pairs]: 1 = the pair's path is row `pair` of paths_dp
    const int8_t *paths_dp;       // [n_pairs][dp_stride]
    int32_t dp_stride;
};

__device__ __forceinline__ const int8_t *path_row(const CommitArgs &a, int pair)
{
    return (a.from_dp && a.from_dp[pair] == 1) ? a.paths_dp + (size_t)pair * (size_t)a.dp_stride : a.paths + (size_t)pair * (size_t)a.path_stride;
}

// grid: n_pairs workgroups of 256 threads: chunk_base[pair][ch] = reference / query columns the path consumes before its 256-element chunk ch.
// Thread t counts chunk g * 256 + t (sixteen 16-byte loads; rows of the path buffer start at multiples of path_stride, hence the byte
// path for rows that are not 16-byte aligned), one workgroup scan per 65536 path elements.
__global__ void __launch_bounds__(256) path_scan_kernel(CommitArgs a)
{
    __shared__ int s_wave[4];
    const int pair = blockIdx.x;
    const int n = a.path_len[pair];
    const int8_t *path = path_row(a, pair);
    const bool aligned = ((size_t)path & 15u) == 0;
    const int nch = (n + 255) >> 8;
    int baseR = 0, baseQ = 0;
    for (int g0 = 0; g0 < nch; g0 += 256) {
        const int ch = g0 + threadIdx.x;
        const int c0 = ch << 8;
        int cR = 0, cQ = 0;
        if (ch < nch) {
            if (aligned && c0 + 256 <= n) {
                const uint4 *w4 = reinterpret_cast<const uint4 *>(path + c0);
                int oddR = 0, oddQ = 0;              // codes 0 / 1 / 2: the reference is consumed unless bit 0 is set, the query unless bit 1 is
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint4 w = w4[k];
                    oddR += __popc(w.x & 0x01010101u) + __popc(w.y & 0x01010101u) + __popc(w.z & 0x01010101u) + __popc(w.w & 0x01010101u);
                    oddQ += __popc(w.x & 0x02020202u) + __popc(w.y & 0x02020202u) + __popc(w.z & 0x02020202u) + __popc(w.w & 0x02020202u);
                }
                cR = 256 - oddR; cQ = 256 - oddQ;
            } else {
                const int e = min(n, c0 + 256);
                for (int c = c0; c < e; ++c) { const int code = path[c]; cR += (code == 0 || code == 2); cQ += (code == 0 || code == 1); }
            }
        }
        int totR, totQ;
        const int exR = baseR + block_scan_int_256(cR, &totR, s_wave);
        const int exQ = baseQ + block_scan_int_256(cQ, &totQ, s_wave);
        if (ch < nch) {
            a.chunk_base[((size_t)pair * a.n_chunks + ch) * 2] = exR;
            a.chunk_base[((size_t)pair * a.n_chunks + ch) * 2 + 1] = exQ;
        }
        baseR += totR;
        baseQ += totQ;
    }
}

// grid: (n_work, ceil(n_chunks / 4)), 256 threads: 1024 columns of the new rows for a group of members of one side, four columns
// (one 32-bit store) per thread.  New row: letter of the old row where the path keeps this side (code 0 or the side's own code), '-'
// elsewhere (alignment-helper.cpp:389-400, 436-447).  Rows start at multiples of the planes' pitch (a multiple of 256): the stores
// are aligned; bytes behind the path's end within the last word are padding of the row's capacity.
__global__ void __launch_bounds__(256) apply_path_kernel(CommitArgs a)
{
    __shared__ int s_wave[4];
    const int32_t *w = a.work + (size_t)blockIdx.x * 3;
    const int side = w[0], pair = side >> 1, isQ = side & 1;
    const int n = a.path_len[pair];
    const int c0 = blockIdx.y * 1024;
    if (c0 >= n) return;
    const int c = c0 + 4 * threadIdx.x;
    const int own = isQ ? 1 : 2;
    const int8_t *path = path_row(a, pair);
    bool keep[4];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pc = path[min(c + k, n - 1)];          // (unconditional: a load under a lane condition is a branch with its own wait)
        const int code = (c + k < n) ? pc : 3;
        keep[k] = (code == 0) | (code == own);
        cnt += keep[k] ? 1 : 0;
    }
    int tot;
    const int src = a.chunk_base[((size_t)pair * a.n_chunks + 4 * blockIdx.y) * 2 + isQ] + block_scan_int_256(cnt, &tot, s_wave);
    if (c >= n) return;
    // the kept columns of this thread are cnt CONSECUTIVE bytes of the old row, from src: one or two aligned 32-bit loads per member
    // (the second only when the bytes straddle a word: it then lies inside the row), spread over the word that is stored;
    // eight members per round so that their loads are in flight together
    const SideDesc sd = a.sides[side];
    const int sh = (src & 3) * 8;
    const bool two = (src & 3) + cnt > 4;
    const size_t lo = (size_t)(src & ~3);
    constexpr int U = 8;
    const int mEnd = w[1] + w[2];
    for (int m = w[1]; m < mEnd; m += U) {
        uint32_t v0[U], v1[U];
        char *to[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mi = sd.member_off + min(m + u, mEnd - 1);
            const size_t off = (size_t)a.member_seq[mi] * a.cap;
            const bool pl = a.member_plane[mi];
            const char *from = (pl ? a.rows1 : a.rows0) + off;
            to[u] = (pl ? a.rows0 : a.rows1) + off;
            v0[u] = cnt ? *reinterpret_cast<const uint32_t *>(from + lo) : 0u;
            v1[u] = two ? *reinterpret_cast<const uint32_t *>(from + lo + 4) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (m + u >= mEnd) break;
            uint64_t v = (((uint64_t)v1[u] << 32) | v0[u]) >> sh;
            uint32_t word = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t ch = keep[k] ? (uint32_t)(v & 0xFFu) : (uint32_t)'-';
                if (keep[k]) v >>= 8;
                word |= ch << (8 * k);
            }
            *reinterpret_cast<uint32_t *>(to[u] + c) = word;
        }
    }
}

// grid: (n_merge, n_chunks), 256 threads: merged cache column j (alignment-helper.cpp:506-539).
template <int P>
__global__ void __launch_bounds__(256) merge_cache_kernel(CommitArgs a)
{
    __shared__ int s_wave[4];
    const int32_t *mg = a.merge + (size_t)blockIdx.x * 4;
    const int pair = mg[0];
    const int n = a.path_len[pair];
    const int c0 = blockIdx.y * 256;
    if (c0 >= n) return;
    const int j = c0 + threadIdx.x;
    const int code = (j < n) ? path_row(a, pair)[j] : 3;
    int tot;
    const int r = a.chunk_base[((size_t)pair * a.n_chunks + blockIdx.y) * 2] + block_scan_256(code == 0 || code == 2, &tot, s_wave);
    const int q = a.chunk_base[((size_t)pair * a.n_chunks + blockIdx.y) * 2 + 1] + block_scan_256(code == 0 || code == 1, &tot, s_wave);
    if (j >= n) return;
    const float *fr = a.cache[mg[1]] + (size_t)r * P;
    const float *fq = a.cache[mg[2]] + (size_t)q * P;
    float *out = a.cache[mg[3]] + (size_t)j * P;
    const float refWeight = a.merge_w[2 * blockIdx.x], qryWeight = a.merge_w[2 * blockIdx.x + 1];
    if (code == 0) {
#pragma unroll
        for (int k = 0; k < P; ++k) out[k] = fr[k] + fq[k];
    } else if (code == 1) {
#pragma unroll
        for (int k = 0; k < P - 1; ++k) out[k] = fq[k];
        out[P - 1] = (float)((double)fq[P - 1] + 1.0 * (double)refWeight);
    } else {
#pragma unroll
        for (int k = 0; k < P - 1; ++k) out[k] = fr[k];
        out[P - 1] = (float)((double)fr[P - 1] + 1.0 * (double)qryWeight);
    }
}

// grid: (n_seqs, n_chunks), 256 threads: current row of every sequence, packed back to back (for the final download).
__global__ void __launch_bounds__(256) gather_rows_kernel(const char *rows0, const char *rows1, int64_t cap, const uint8_t *plane,
                                                          const int32_t *len, const int64_t *off, char *out)
{
    const int s = blockIdx.x;
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= len[s]) return;
    out[off[s] + c] = (plane[s] ? rows1 : rows0)[(size_t)s * cap + c];
}

// The same for a list of sequences (ids[t] -> out + off[t]), and its inverse: rows of other ranks' subtrees arriving in this store (twl_store_read_rows_of /
// twl_store_write_rows: the one exchange of a sharded run whose subtrees were aligned by their owners alone).  grid: (n_ids, n_chunks), 256 threads.
__global__ void __launch_bounds__(256) gather_rows_of_kernel(const char *rows0, const char *rows1, int64_t cap, const uint8_t *plane, const int32_t *ids,
                                                             const int32_t *len, const int64_t *off, char *out)
{
    const int t = blockIdx.x, s = ids[t];
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= len[t]) return;
    out[off[t] + c] = (plane[s] ? rows1 : rows0)[(size_t)s * cap + c];
}
__global__ void __launch_bounds__(256) scatter_rows_of_kernel(char *rows0, char *rows1, int64_t cap, const uint8_t *plane, const int32_t *ids,
                                                              const int32_t *len, const int64_t *off, const char *in)
{
    const int t = blockIdx.x, s = ids[t];
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= len[t]) return;
    (plane[s] ? rows1 : rows0)[(size_t)s * cap + c] = in[off[t] + c];
}

}  // namespace twl

// twilight_amd/csrc/talco_nuc.hip.h -- the nucleotide (P = 6) TALCO-XDrop kernel of round 2: same results as
// talco_kernel<6,...> (talco_kernel.hip.h), a diagonal step built for instruction count.
//
// What it computes: Talco_xdrop::Align_freq / Tile / Traceback of the reference CPU path
// (/root/reference/src/TALCO-XDrop.cpp:62-108, :233-689, :134-231), fp32, x86 TALCO_SIMD operation order.
//
// Why a second kernel.  Measured on MI355X (tools/micro/issue_rates*.hip, profiles/r02/issue_rates*.log): one wave issues one
// instruction of ANY kind per ~4 cycles; a CU retires at most ~1 scalar-type instruction (s_*, v_readlane, v_writelane) per
// cycle; an LDS round trip is ~50 cycles and an s_barrier of 16 waves ~20.  The round-1 kernel issued ~800 instructions per
// wave and diagonal (of which ~340 scalar and ~70 SGPR spill moves), so a lone pair's diagonal cost 1.6 us and a full CU was
// bound by the scalar unit.  The diagonal step is therefore rebuilt around "fewer instructions per wave":
//   * three loops, one per phase of a tile (A: k < marker-1, traceback pointers only; B: the two marker diagonals; C: k > marker,
//     convergence pointers only) instead of per-step phase tests;
//   * bands kept as (low, width-1) pairs that rotate, one unsigned compare per band test, one combined stop test per step
//     with the exact decoding out of line; the watchdog counts tiles' diagonals, not steps;
//   * per-diagonal reductions without reset or rotation: the running maximum is an LDS float max that only lanes above the
//     current maximum touch; first/last unpruned row are ONE two-lane ds_max_u32 on (diagonal tag << 16 | row) words, so a
//     stale slot loses by construction; slots alternate by parity;
//   * the running maximum / X-drop threshold live in (uniform) vector registers: no scalar float emulation through integer keys;
//   * q[m]*M[l][m] (first rounding of (q*M)*r, :386) kept per row for the whole tile; the IEEE division by refNum*qryNum
//     (:444) as the hoisted-reciprocal form of the same correctly rounded sequence (see fast_div below);
//   * the traceback word of a row block is stored every 8th diagonal unconditionally (no dirty tracking).
// Everything observable (band evolution, tile boundaries, paths, error codes, band-cell counts) is unchanged and is checked
// against the oracle by the same parity suite.
#pragma once
#include "talco_kernel.hip.h"

#if defined(TWL_EXP_NOPRIO)      // experiment builds: no wave priorities
#define TWL_SETPRIO(x) do {} while (0)
#else
#define TWL_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif

namespace twl {

struct NArgs {
    const float *cols;        // packed columns [pair][2][seq_len][P+2]: f0..f(P-1), gapOpen, gapExtend (32 B for P = 6, 96 B for P = 22)
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
    int32_t n_pairs_total;
    int32_t step_slack;       // watchdog: a pair may run at most (R+Q+2)*((R+Q)/(marker-1)+4) + step_slack diagonals in total
    float gap_open, gap_extend, gap_char;
    const uint8_t *gc_zero;   // optional [pair]: 1 = this pair's gapCharScore is 0 whatever gap_char says (alignment-cpu.cpp:88 decides per pair)
    int32_t xdrop, flen, marker;
    float M[25];              // nucleotide scoreMatrix[l][m] row-major 5x5 (the protein matrix comes through M24)
    const float *M24;         // protein: [21][24] matrix rows padded to 24 floats (device memory)
    // protein, few pairs (matrix mode 4): column scores precomputed by score_matrix_kernel, diagonal-major per pair:
    // sim[sim_off[pair] + (i + j) * pitch + i] with i = query row, j = reference column, pitch = (Q + 63) & ~63
    const float *sim;
    const long long *sim_off;
    unsigned long long *team; // speculative tile start (SPEC kernels): [n_items][16] mailbox words, zeroed by the host
    float *simdump;           // DUMP kernels (diagnostics, one pair): [Q][R] column scores as the DP evaluated them
    // tile-parallel alignment (MT kernels, see below)
    const int32_t *mt_jobs;   // [n_items][3] {pair, slot, row of the pair in the mt_* tables} (MT 1, 2; `items` is unused there)
    int32_t *mt_chain;        // [row][mt_slots][2]  (row = the pair's position in the stitch launch) predicted start {ref_idx, qry_idx} of tile `slot`; ref_idx < 0: unknown
    int32_t *mt_rec;          // [row][mt_slots][kMtRec] result of the tile that was run from the predicted start
    int8_t *mt_seg;           // [row][mt_slots][mt_segcap] its path segment, forward order
    int32_t *mt_spath;        // [row][mt_sp_pitch] scouts: query row of the path cell on anti-diagonal d; -1 = the path skips d; < -1 = unknown
    unsigned long long *mt_stat;   // [4] {tiles taken from the records, tiles run in line, scouts that failed, -}
    int32_t *mt_front;        // [row][8] how far the stitch launches have come: {0 untouched / 1 suspended at a tile without a record / 2 done, tile, ref_idx, qry_idx, pos, -, cells lo, cells hi}
    int32_t mt_slots, mt_segcap, mt_sp_pitch, mt_lead, mt_marg;
    const int32_t *mt_anchor; // [row][mt_slots] scouts: 128 + the offset of q - r at which the consensus letters of the two profiles agree, trusted; < 0: none (nullptr: no anchors at all)
    int32_t mt_lead2;         // anti-diagonals an anchored scout starts ahead of its tile boundary
    int32_t mt_inline;        // MT 3: 1 = a tile without a matching record is computed in line (last round); 0 = the pair is suspended there
};

typedef float nuc_f4 __attribute__((ext_vector_type(4)));
typedef float nuc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 lds_ld128(unsigned addr)      // one ds_read_b128 at a byte address kept in a register
{
    const nuc_f4 v = *reinterpret_cast<const __attribute__((address_space(3))) nuc_f4 *>((const __attribute__((address_space(3))) char *)nullptr + addr);
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef int nuc_i2 __attribute__((ext_vector_type(2)));
typedef int nuc_i4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T lds_ld(unsigned addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) T *>((const __attribute__((address_space(3))) char *)nullptr + addr);
}
template <class T>
__device__ __forceinline__ void lds_st(unsigned addr, T v)
{
    *reinterpret_cast<__attribute__((address_space(3))) T *>((__attribute__((address_space(3))) char *)nullptr + addr) = v;
}
template <int OFF>
__device__ __forceinline__ void lds_max_u32_off(unsigned addr, unsigned v) { asm volatile("ds_max_u32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_max_f32_off(unsigned addr, float v) { asm volatile("ds_max_f32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
// ds atomics on (address register + immediate offset)
#define TWL_LDS_ATOM(NAME)                                                                                                     \
    template <int OFF>                                                                                                         \
    __device__ __forceinline__ void NAME##_off(unsigned addr, int v) { asm volatile(#NAME " %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
TWL_LDS_ATOM(ds_min_i32)
TWL_LDS_ATOM(ds_max_i32)
TWL_LDS_ATOM(ds_or_b32)
#undef TWL_LDS_ATOM
// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter, which
// makes every eighth diagonal wait for the round trip of its traceback store; those stores are only read after the tile's
// final __syncthreads().
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_max_u32(unsigned addr, unsigned v) { asm volatile("ds_max_u32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_max_f32(unsigned addr, float v) { asm volatile("ds_max_f32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// n / d for a divisor that is fixed for the whole pair.  `r` is the refined reciprocal fma(fma(-d, rcp(d), 1), rcp(d), rcp(d)).
// This is the instruction sequence the compiler emits for an IEEE fp32 division (v_div_scale, v_rcp, 4 fma, v_div_fmas,
// v_div_fixup) with the parts that depend on d alone hoisted and the scaling steps left out.  v_div_scale rescales only when
// d or n/d is near the ends of the exponent range or |n| < 2^-103, and v_div_fixup only patches zeros, infinities and NaNs,
// so for 1 <= d <= 2^40 and n == 0 or 2^-73 <= |n| <= 2^80 both give the same bits (a zero quotient may differ in sign, which
// no comparison or sum downstream can see).  The kernel guards that range: profile entries and scores outside it send the
// pair to the IEEE-division kernel (err = kErrOverflow, re-run by the next stage of the chain).  tests/test_gpu_parity.py
// compares the two forms bit for bit on device (twl_debug_fast_div).
__device__ __forceinline__ float fast_div(float n, float d, float r)
{
    const float q0 = n * r;
    const float t0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(t0, r, q0);
    const float t1 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(t1, r, q1);
}
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
// a profile entry fast_div's guard accepts: zero, or 2^-20 <= |x| <= 2^30
__device__ __forceinline__ bool div_guard_bad(float x)
{
    const float ax = __builtin_fabsf(x);
    return (x != 0.0f) && !(ax >= 9.5367431640625e-07f && ax <= 1073741824.0f);
}

// ---- speculative tile start (SPEC = true) ----
// A tile runs ~1450 diagonals: 1024 up to the marker and ~420 more until every surviving cell agrees on the cell of the marker
// diagonal all paths run through (TALCO-XDrop.cpp:585-612); the next tile starts there with scores at zero (:615-655, :77-106), so it
// depends on the POSITION of that cell only.  With fewer pairs than compute units, two workgroups work on one pair: they take the tiles
// in turn, and the one whose turn is next starts as soon as the other reaches its marker, from the best-scoring cell of the marker
// diagonals (the guess).  When the running tile has converged its workgroup publishes the true position: the partner carries on if the
// guess was right and restarts from the true position otherwise -- a wrong guess costs nothing but the idle workgroup's time, and
// the result is the same either way.  Mailbox words (global memory, one 8-byte agent-scope atomic each, tag = tile + 1):
//   guess[t & 3], truth[t & 3] = tag:16 | flags:16 | ref_idx:16 | qry_idx:16      pos[t & 3] = tag:16 | 0:16 | output offset:32
constexpr unsigned kTeamLast = 1u, kTeamErr = 2u;
constexpr int kTeamWords = 16, kTeamGuess = 0, kTeamTruth = 4, kTeamPos = 8, kTeamStat = 12;
__device__ __forceinline__ unsigned long long team_word(unsigned tag, unsigned flags, unsigned hi, unsigned lo)
{
    return ((unsigned long long)(tag & 0xFFFFu) << 48) | ((unsigned long long)(flags & 0xFFFFu) << 32) | ((unsigned long long)(hi & 0xFFFFu) << 16) | (lo & 0xFFFFu);
}
__device__ __forceinline__ unsigned long long team_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void team_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane waits (bounded: ~2 s) until the word carries the tag; 0 = gave up: the partner is not co-resident or far too slow (a GPU
// shared with another process, a profiler serialising the workgroups).  That is a scheduling condition, not an alignment error: the
// caller reports the internal re-run code (kErrOverflow), the host re-runs the pair on the plain kernel of the next stage
__device__ __forceinline__ unsigned long long team_wait(const unsigned long long *p, unsigned tag, const unsigned long long *alt = nullptr)
{
    for (int spin = 0; spin < (1 << 21); ++spin) {
        const unsigned long long v = team_load(p);
        if ((unsigned)(v >> 48) == tag) return v;
        if (alt) { const unsigned long long u = team_load(alt); if ((unsigned)(u >> 48) == tag) return u | (1ull << 47); }      // bit 47: "this is the guess"
        __builtin_amdgcn_s_sleep(16);
    }
    return 0ull;
}
// order-preserving key of a score for an integer maximum (the guess only: never part of a result)
__device__ __forceinline__ unsigned score_key(float f)
{
    const unsigned b = (unsigned)__float_as_int(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ---- tile-parallel alignment (MT != 0) ----
// A tile is a pure function of its start cell (TALCO-XDrop.cpp:77-93: Tile() reads reference_idx / query_idx and `tile == 0`, nothing else
// of the previous tile), and every tile after the first advances ref_idx + qry_idx by marker or marker - 1 (:616-619).  So if the start
// cells of all tiles of a pair were known, the tiles could run side by side on as many compute units.  They can be PREDICTED: a DP that
// starts a few hundred anti-diagonals before a tile boundary, from a cell merely NEAR the optimal path, runs into that path (the band
// opens by a row per diagonal, the X-drop keeps everything within `xdrop` of the best cell) and from there on its traceback path IS the
// optimal path.  Four launches per level:
//   MT 2  scouts   one workgroup per (pair, tile boundary t): a short tile from the cell of anti-diagonal (marker-1)*t - 1 - lead that lies on
//                  the straight line between the corners, stopped right after its own marker (placed mt_marg diagonals behind the boundary
//                  range), traced back from the better of the best cells of its two marker diagonals; writes where its path crosses the
//                  anti-diagonals [(marker-1)*t - 1, marker*t + 1] into mt_spath
//   MT 4  pair scout  one workgroup per pair (wide re-runs, round 4): the DP from (0, 0) to the last anti-diagonal with every traceback word kept, no
//                  marker and no convergence test; its path from the end cell gives mt_spath for ALL anti-diagonals.  For pairs whose tiles do not
//                  converge (diffuse profiles: every tile runs to the end of the pair and its successor starts where the path from the END cell
//                  crosses the marker diagonal) a local scout cannot know that path; the suffix of the pair's global path does
//   (mt_chain_kernel: walks mt_spath from (0, 0): start of tile t = the path cell on diagonal s + marker, or on s + marker - 1 when the
//                  path steps over that diagonal -- the reference's state 3, :520-524)
//   MT 1  tiles    one workgroup per (pair, tile): the tile from the predicted start, result + path segment into mt_rec / mt_seg
//   MT 3  stitch   one workgroup per pair runs the ordinary tile loop; a tile whose TRUE start equals the start a record was computed
//                  from takes the record (same function, same argument), any other tile is computed in line as always.
// A wrong prediction that also gets the anti-diagonal wrong (marker vs marker - 1) would shift every later boundary of the pair, so the
// chain / tiles / stitch launches run in ROUNDS: a stitch launch that is not the last one suspends a pair at the first tile without a
// matching record (mt_front), the next chain launch re-derives the pair's remaining starts from that TRUE cell, the next tile launch runs
// the tiles whose record does not match the new prediction; the last stitch launch computes whatever is still missing in line.
// Results are those of the plain loop by construction; predictions only decide how much of it is already done.
constexpr int kMtRec = 12;     // {1 = valid / 2 = the job failed, start ref, start qry, next ref, next qry, last_tile, segment bytes, tail dir, tail len, band cells, error code of a failed job, its window rows}

// start cells of all tiles of a pair from the scouts' path samples (one thread per pair)
__global__ void mt_chain_kernel(const int32_t *spath, int sp_pitch, const int32_t *len, const int32_t *items, int n_items, int32_t *chain, int slots,
                                int marker, int perturb, const int32_t *front)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= n_items) return;
    const int pair = items[it];
    const int R = len[2 * pair], Q = len[2 * pair + 1];
    // (the tables of a tile-parallel level are indexed by the pair's position in the launch, not by its id: they cost memory and fills for the pairs that run only)
    const int32_t *sp = spath + (size_t)it * (size_t)sp_pitch;
    int32_t *ch = chain + (size_t)it * (size_t)slots * 2;
    const int32_t *fr = front + (size_t)it * 8;
    if (fr[0] == 2) return;                       // the pair is finished
    int s = 0, t = 1;
    if (fr[0] == 1) {                             // later rounds: from the true start of the first tile without a record
        t = fr[1];
        if (t >= slots) return;
        ch[2 * t] = fr[2]; ch[2 * t + 1] = fr[3];
        s = fr[2] + fr[3];
        t += 1;
    } else { ch[0] = 0; ch[1] = 0; }
    for (; t < slots; ++t) {
        const int d = s + marker;
        if (d > R + Q - 2) break;
        int q = sp[d];
        if (q >= 0) s = d;
        else if (q == -1 && d - 1 > s && sp[d - 1] >= 0) { q = sp[d - 1]; s = d - 1; }
        else break;
        int r = s - q;
        // development / test aid: spoil every perturb-th prediction, so that the stitch kernel has tiles to compute in line
        if (perturb > 0 && (t % perturb) == 0) { if (q > 0 && r + 1 < R) { q -= 1; r += 1; } else if (r > 0 && q + 1 < Q) { q += 1; r -= 1; } }
        if (r < 0 || r >= R || q >= Q) break;
        ch[2 * t] = r; ch[2 * t + 1] = q;
    }
    for (; t < slots; ++t) { ch[2 * t] = -1; ch[2 * t + 1] = -1; }
}

// ---- anchored scouts (round 5) ----
// A scout needs its long lead (320 anti-diagonals) because the straight line between the corners is up to ~55 rows off the path on the families
// measured: the band has to open that far before the scout can run into the path.  Where the path IS can be read off the profiles before any DP: around
// the cell of the straight line the consensus letters of the two profiles agree on one diagonal offset and on no other (tests/study/tile_predict_study.c:
// +-32 columns, offsets -128..127).  mt_anchor_kernel finds that offset for every tile boundary; a scout whose anchor is trusted -- or lies between two
// trusted ones that agree (the path drifts by a few rows per tile; an offset that jumps is a long indel and says nothing about its neighbours) -- starts
// from the anchored cell only mt_lead2 (96) anti-diagonals ahead.  On the 154 pairs of levels 15-31 of 10 000 x 10 kbp (the CPU study, with the nine windows of
// mt_anchor_kernel): all but 2 of 3080 boundaries anchored, 3078 true starts found against 3079 with the long lead, 61 % fewer scout diagonals; on the device the scouts of a
// pass 39 -> 15 ms.  Predictions only: a wrong anchor costs what any wrong prediction costs -- a second round of its level.
constexpr int kAnchorHalf = 32, kAnchorOffsets = 256, kAnchorNear = 3, kAnchorJump = 12;
constexpr int kAnchorMinMatches = 36, kAnchorMinGap = 8;      // trusted: >= 55 % of the 64 columns agree and the best other offset (not a neighbour) is >= 12 % behind

// the cell of anti-diagonal d0 on the straight line between the corners (false: none)
__device__ __forceinline__ bool scout_line_cell(int d0, int R, int Q, int &r, int &q)
{
    q = (int)(((long long)d0 * Q) / (R + Q));
    q = min(q, Q - 1);
    r = d0 - q;
    if (r > R - 1) { r = R - 1; q = d0 - r; }
    return !(q > Q - 1 || r < 0);
}
// the anchor of boundary `slot` of a pair from the table of its row: its own when trusted, else between two trusted neighbours that agree
__device__ __forceinline__ bool scout_anchor(const int32_t *an, int slot, int slots, int &o)
{
    const int w = __builtin_amdgcn_readfirstlane(an[slot]);
    if (w >= 0) { o = w - 128; return true; }
    int l = -1, r = -1, ol = 0, orr = 0;
    for (int u = 1; u <= kAnchorNear; ++u) {
        if (l < 0 && slot - u >= 1) { const int v = __builtin_amdgcn_readfirstlane(an[slot - u]); if (v >= 0) { l = slot - u; ol = v - 128; } }
        if (r < 0 && slot + u < slots) { const int v = __builtin_amdgcn_readfirstlane(an[slot + u]); if (v >= 0) { r = slot + u; orr = v - 128; } }
    }
    if (l < 0 || r < 0 || abs(orr - ol) > kAnchorJump * (r - l)) return false;
    o = ol + (orr - ol) * (slot - l) / (r - l);
    return true;
}
// the anchored start cell on anti-diagonal d0: q - r moved by o (rounded away from zero to an even step)
__device__ __forceinline__ bool scout_anchored_cell(int d0, int R, int Q, int o, int &r, int &q)
{
    int r0, q0;
    if (!scout_line_cell(d0, R, Q, r0, q0)) return false;
    q = q0 + (o >= 0 ? (o + 1) / 2 : -((-o + 1) / 2));
    r = d0 - q;
    if (q < 0) { q = 0; r = d0; }
    if (r < 0) { r = 0; q = d0; }
    return q < Q && r < R;
}

// one workgroup of 256 threads per scout job {pair, slot, row}: thread t counts the columns i in [-32, 32) with consensus(ref, r0 + i) == consensus(qry, q0 + i + t - 128),
// first in the window around the straight line's cell, then -- until one is trusted -- in windows moved along the diagonal by +24, -24, +48, ... -96 columns (the path's
// offset is the same a few columns on unless an indel lies between; the top 52 pairs of 10 000 x 10 kbp: 18 % of the boundaries without an anchor with the one window, 0.2 % with the nine)
constexpr int kAnchorShift = 24, kAnchorShifts = 4, kAnchorPad = kAnchorShift * kAnchorShifts;
template <int P>
__global__ __launch_bounds__(256) void mt_anchor_kernel(const float *cols, const int32_t *len, int seq_len, const int32_t *jobs, int n_jobs, int32_t *anchor, int slots, int marker, int lead2)
{
    static_assert(kAnchorOffsets == 256 && 2 * kAnchorHalf + 2 * kAnchorPad == 256, "one thread per offset, one per staged reference letter");
    __shared__ unsigned char s_r[2 * kAnchorHalf + 2 * kAnchorPad], s_q[2 * kAnchorHalf + 2 * kAnchorPad + kAnchorOffsets];
    __shared__ int s_cnt[kAnchorOffsets], s_done;
    const int job = blockIdx.x;
    if (job >= n_jobs) return;
    const int pair = jobs[3 * job], slot = jobs[3 * job + 1], row = jobs[3 * job + 2];
    const int R = len[2 * pair], Q = len[2 * pair + 1];
    const int spLo = (marker - 1) * slot - 1;
    const int d0 = max(spLo - lead2, 0);
    int r0, q0;
    if (spLo > R + Q - 2 || R < 2 || Q < 2 || !scout_line_cell(d0, R, Q, r0, q0)) return;      // (the table is filled with -1)
    // consensus letter of a column: the most frequent of A, C, G, T (of the twenty amino acids) when it is at least as frequent as the gap; else a letter that matches nothing
    auto letter = [&](int side, int col, int n) -> unsigned char {
        if (col < 0 || col >= n) return (unsigned char)(100 + side);
        const float *pf = cols + (((size_t)pair * 2 + side) * (size_t)seq_len + col) * (P + 2);
        int best = 0; float bc = pf[0];
        for (int j = 1; j < P - 2; ++j) if (pf[j] > bc) { bc = pf[j]; best = j; }
        return (bc > 0.0f && bc >= pf[P - 1]) ? (unsigned char)best : (unsigned char)(100 + side);
    };
    const int t = threadIdx.x;
    s_r[t] = letter(0, r0 - kAnchorHalf - kAnchorPad + t, R);
    for (int u = t; u < 2 * kAnchorHalf + 2 * kAnchorPad + kAnchorOffsets; u += 256) s_q[u] = letter(1, q0 - kAnchorHalf - kAnchorPad - kAnchorOffsets / 2 + u, Q);
    if (t == 0) s_done = 0;
    __syncthreads();
    for (int sh = 0; sh <= 2 * kAnchorShifts; ++sh) {
        const int base = kAnchorPad + ((sh == 0) ? 0 : ((sh & 1) ? kAnchorShift * ((sh + 1) / 2) : -kAnchorShift * (sh / 2)));
        int c = 0;
        for (int i = 0; i < 2 * kAnchorHalf; ++i) c += (s_r[base + i] == s_q[base + i + t]) ? 1 : 0;      // offset o = t - 128
        s_cnt[t] = c;
        __syncthreads();
        if (t < 64) {
            // the best offset: most matches, ties to the smallest |o|, the negative one first (the order 0, -1, 1, -2, 2, ... of the study)
            int key = -1;
            for (int u = t; u < kAnchorOffsets; u += 64) {
                const int o = u - kAnchorOffsets / 2;
                const int rank = (o == 0) ? 0 : (o < 0 ? -2 * o - 1 : 2 * o);
                key = max(key, (s_cnt[u] << 16) | (0xFFFF - rank));
            }
            for (int m = 32; m >= 1; m >>= 1) key = max(key, __shfl_xor(key, m, 64));
            const int rank = 0xFFFF - (key & 0xFFFF);
            const int bo = (rank == 0) ? 0 : ((rank & 1) ? -(rank + 1) / 2 : rank / 2);
            int second = -1;
            for (int u = t; u < kAnchorOffsets; u += 64) if (abs(u - kAnchorOffsets / 2 - bo) > 2) second = max(second, s_cnt[u]);
            for (int m = 32; m >= 1; m >>= 1) second = max(second, __shfl_xor(second, m, 64));
            const int bc = key >> 16;
            if (t == 0 && bc >= kAnchorMinMatches && bc - second >= kAnchorMinGap) { anchor[(size_t)row * slots + slot] = 128 + bo; s_done = 1; }
        }
        __syncthreads();
        if (s_done) break;
    }
}

template <int W, int RPL>
struct NCfg {
    static constexpr int NV = W * RPL;          // 64-row blocks resident at once
    static constexpr int WINDOW = 64 * NV;      // rows
    static constexpr int NB = NV + 2;           // ring blocks
    static constexpr int CAP = 64 * NB;         // ring columns
    static constexpr int THREADS = 64 * W;
};

// P = 6 (nucleotide) or 22 (protein).  MM = column-score mode, host-selected (every mode of an alphabet computes the same sums, see
// talco_kernel): nucleotide 0 general 5x5, 1 zero N row/column (4x4 core), 2 mode 1 with the match / transition / transversion
// structure (three products per row letter), 5 modes 1 / 2 for query rows with one non-zero letter (single sequences); protein 3 loop over the non-zero letters of the reference column, 4 scores
// precomputed by score_matrix_kernel for the whole R x Q matrix (launches with few pairs: the other CUs are idle anyway).
// SP (round 5) = what the host knows about every pair of the launch, taken out of the per-block tests (15 of the ~111 instructions of a leaf x leaf block step are
// the tests "does a row of this block hold gap letters", "does a column in the band", "is the denominator 1" and their branches; profiles/r05/isa_block_step.json):
//   0  nothing: the tests run (every other launch)
//   1  leaf x leaf: single sequences on both sides -- no gap letter anywhere and refNum * qryNum = 1: neither the gap-letter terms (:394-395) nor the division (:444)
//      exist in the step (leaf level of 10 000 x 10 kbp 94.7 -> 88.6 ms).  The promise is checked: a gap letter or another denominator sends the pair to the
//      general kernel (kErrGuard).  (The opposite specialisation -- several sequences on both sides, terms always formed -- was built and measured: no level of a
//      random tree qualifies, every level joins some leaf to a subtree.)
template <int P, int W, int RPL, int MM, int MINW, bool SPEC = false, bool DUMP = false, int MT = 0, int SP = 0>
__global__ __launch_bounds__(64 * W, MINW) void talco_lean_kernel(NArgs a)
{
    static_assert(SP == 0 || (SP == 1 && P == 6 && !SPEC && !DUMP && MM == 5), "the leaf x leaf step: nucleotide, one-letter query rows");
    static_assert((P == 6 && ((MM >= 0 && MM <= 2) || MM == 5)) || (P == 22 && (MM == 3 || MM == 4)), "profile width / column-score mode");
    static_assert(MT == 0 || (!SPEC && !DUMP), "the tile-parallel kernels are plain ones");
    constexpr bool GUESS = SPEC || MT == 2;      // the best cells of the two marker diagonals are collected
    using C = NCfg<W, RPL>;
    constexpr int NV = C::NV, WINDOW = C::WINDOW, NB = C::NB, CAP = C::CAP;
    // first products q[m]*M[l][m] kept per row as PAIRS over two matrix rows, {l = 2h, l = 2h+1}, so that the second product and the
    // sums of two rows are one packed instruction each with every operand in an aligned register pair: NQP pairs (+ row 4 in mode 0)
    constexpr int NQP = (MM == 2 || MM == 1) ? 8 : (MM == 0 ? 10 : (MM == 5 ? 2 : 1));
    constexpr int NQM = (MM == 0) ? 5 : 1;
    // Geometries of three or more blocks per wave (the wide windows) do not keep the first products per row: sixteen registers per block put
    // the 16-wave x 3-block kernel 60 registers over its budget of 128 (212-244 B of scratch per lane, reloaded on the diagonal's critical
    // path: 2.6-3.8 us per diagonal of a 2300-row band).  They form q[m]*M[l][m] again in every cell -- the same first rounding (:386), eight packed multiplies more.
#if defined(TWL_EXP_QPRE)      // experiment builds: force either form for every geometry
    constexpr bool QPRE = !(MM == 2) || TWL_EXP_QPRE;
#else
    constexpr bool QPRE = !((RPL >= 3 || MINW >= 5) && MM == 2);      // (five waves per SIMD: 96 registers)
#endif
    constexpr int F4 = (P + 2) / 4;                       // float4 per packed column: 2 or 6
    constexpr bool SPARSE = (MM == 3), PRESIM = (MM == 4);
    constexpr int RP = PRESIM ? 1 : F4;                  // ring planes: presim keeps only {X letter, gap, gapOpen, gapExtend}
    constexpr int QN = PRESIM ? 1 : P;                   // query letters kept per row

    __shared__ float4 s_ring[RP * CAP];       // plane-major: plane t of a column = its floats 4t .. 4t+3; the last plane = {.., gap letter, gapOpen, gapExtend}
    __shared__ uint32_t s_rmask[SPARSE ? CAP : 1];        // protein sparse: non-zero-letter bitmask of every ring column
    __shared__ float4 s_M4[SPARSE ? 21 * 6 : 1];          // protein sparse: matrix rows padded to 24 floats
    // everything that alternates with the parity of the diagonal sits in one struct per parity, so that ONE register
    // (vcur / vprev: the LDS address of the current / previous diagonal's struct) selects it and the rest is an immediate offset
    struct ParBuf {
        int cd[WINDOW + 4];          // offset-addressed mirror of the reference's CD rows (see talco_kernel)
        int4 exch[NV];               // mailbox: lane 63 of a block -> lane 0 of the next block {S, I, CS, CI}
        unsigned red[4];             // {running max S (float bits), low-end tag, high-end tag, -}
        int conv[4];                 // {vmin, vmax, flags, -}
        int edge[2 * NV];            // phase C: CS of the first / last unpruned row of every block (the cheap pre-test of the convergence test)
        int4 trash[64];              // per-lane trash slots: single-lane LDS side effects without touching EXEC
    };
    __shared__ ParBuf s_par[2];
    __shared__ int s_misc[8];      // queue item, CS[last_k][0], err, pos of the traceback wave; [4] some wave met an entry outside fast_div's range
    __shared__ int8_t s_rev[2 * kMaxMarker + 16];
    __shared__ unsigned long long s_team[4];      // SPEC: {broadcast word, decision of the last poll, best cell of diagonal marker-1, of diagonal marker}
    constexpr unsigned O_CD = (unsigned)offsetof(ParBuf, cd), O_EXCH = (unsigned)offsetof(ParBuf, exch), O_RED = (unsigned)offsetof(ParBuf, red),
                       O_CONV = (unsigned)offsetof(ParBuf, conv), O_TRASH = (unsigned)offsetof(ParBuf, trash), O_EDGE = (unsigned)offsetof(ParBuf, edge);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *tb = a.tb + (size_t)blockIdx.x * (size_t)a.tb_words;
    const float inf = (float)(2.0 * (double)a.xdrop + 1.0);     // TALCO-XDrop.cpp:252
    // (uniform, and wanted as a scalar operand: in a vector register it was the value the 96-register kernels spilled onto the diagonal's path)
    const float xdropf = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)a.xdrop)));
    if constexpr (SPARSE) {
        for (int t = threadIdx.x; t < 21 * 6; t += C::THREADS) s_M4[t] = reinterpret_cast<const float4 *>(a.M24)[t];
    }

    int specRound = 0;
    for (;;) {
        int item;
        if constexpr (SPEC) {         // two workgroups per pair, no queue: workgroups 2t and 2t+1 are the team of item t
            if (specRound++) break;
            item = (int)(blockIdx.x >> 1);
            __syncthreads();
        } else {
            if (threadIdx.x == 0) s_misc[0] = atomicAdd(a.queue, 1);
            __syncthreads();
            item = __builtin_amdgcn_readfirstlane(s_misc[0]);
        }
        if (item >= a.n_items) break;
        const int pair = __builtin_amdgcn_readfirstlane((MT == 1 || MT == 2) ? a.mt_jobs[3 * item] : a.items[item]);
        const int slot = (MT == 1 || MT == 2) ? __builtin_amdgcn_readfirstlane(a.mt_jobs[3 * item + 1]) : 0;
        const int mtx = (MT == 1 || MT == 2) ? __builtin_amdgcn_readfirstlane(a.mt_jobs[3 * item + 2]) : item;      // row of this pair in the mt_* tables: its position in the stitch launch
        const int R = a.len[2 * pair], Q = a.len[2 * pair + 1];
        const float gc = (a.gc_zero && __builtin_amdgcn_readfirstlane((int)a.gc_zero[pair])) ? 0.0f : a.gap_char;
        const bool gcNZ = (gc != 0.0f);
        const unsigned long long gcMask = gcNZ ? ~0ull : 0ull;
        // MT 2 (scout of tile boundary `slot`): the anti-diagonals it reports and its own marker behind them
        const int spLo = (a.marker - 1) * slot - 1, spHi = a.marker * slot + 1;
        // ... from the cell the profiles' consensus letters point at, a short lead ahead, when there is one (mt_anchor_kernel); else from the straight line
        int anchRef = -1, anchQry = -1;
        bool anchored = false;
        if constexpr (MT == 2) {
            if (a.mt_anchor && !(spLo > R + Q - 2 || R < 2 || Q < 2)) {
                int o = 0;
                if (scout_anchor(a.mt_anchor + (size_t)mtx * a.mt_slots, slot, a.mt_slots, o)) anchored = scout_anchored_cell(max(spLo - a.mt_lead2, 0), R, Q, o, anchRef, anchQry);
            }
        }
        const int scoutD0 = max(spLo - (anchored ? a.mt_lead2 : a.mt_lead), 0);
        // MT 4 (the scout of a whole pair): no marker at all -- phase A to the last anti-diagonal, every traceback word kept
        const int marker = (MT == 2) ? min(spHi + a.mt_marg - scoutD0, kMaxMarker) : ((MT == 4) ? R + Q + 8 : a.marker);
        const float denom = (float)a.num[2 * pair] * (float)a.num[2 * pair + 1];   // :255,:269
        const bool denomOne = (denom == 1.0f);
        const float rden = refined_rcp(denom);
        const float4 *colsR = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 0) * (size_t)a.seq_len * (P + 2));
        const float4 *colsQ = reinterpret_cast<const float4 *>(a.cols + ((size_t)pair * 2 + 1) * (size_t)a.seq_len * (P + 2));
        const float *simP = PRESIM ? a.sim + a.sim_off[pair] : nullptr;
        const int simPitch = (Q + 63) & ~63;

        // NOTE on control flow (as in talco_kernel): no `continue`, and every single-lane block is followed by a workgroup
        // barrier before a loop back-edge.
        int ref_idx = 0, qry_idx = 0, tile = 0, pos = 0, err = 0;
        bool last_tile = (R <= 0 || Q <= 0);
        if constexpr (MT == 1) {          // one tile, from its predicted start (tile 0: from the corner)
            if (slot > 0) {
                ref_idx = __builtin_amdgcn_readfirstlane(a.mt_chain[((size_t)mtx * a.mt_slots + slot) * 2]);
                qry_idx = __builtin_amdgcn_readfirstlane(a.mt_chain[((size_t)mtx * a.mt_slots + slot) * 2 + 1]);
                tile = 1;
                if (ref_idx < 0) last_tile = true;      // no prediction for this tile
            }
            // later rounds: only the tiles the stitch launches have not passed yet, and only where no record of the predicted start exists
            const int32_t *fr = a.mt_front + (size_t)mtx * 8;
            const int fs = __builtin_amdgcn_readfirstlane(fr[0]), ft = __builtin_amdgcn_readfirstlane(fr[1]);
            const int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + slot) * kMtRec;
            const int rv = __builtin_amdgcn_readfirstlane(rc[0]), rr = __builtin_amdgcn_readfirstlane(rc[1]), rq = __builtin_amdgcn_readfirstlane(rc[2]);
            if (fs == 2 || (fs == 1 && slot < ft) || (rv == 1 && rr == ref_idx && rq == qry_idx)) last_tile = true;
        }
        if constexpr (MT == 2) {          // the cell of diagonal scoutD0 on the straight line between the corners
            tile = 1;
            if (spLo > R + Q - 2 || R < 2 || Q < 2) last_tile = true;
            else if (anchored) { ref_idx = anchRef; qry_idx = anchQry; }
            else if (!scout_line_cell(scoutD0, R, Q, ref_idx, qry_idx)) last_tile = true;
        }
        if constexpr (MT == 4) {          // the row tags hold k + 1 in 16 bits; every group of 8 diagonals needs its traceback words
            if (R < 2 || Q < 2 || R + Q > 65000 || (size_t)(((R + Q) >> 3) + 2) * (size_t)WINDOW > (size_t)a.tb_words) last_tile = true;
        }
        const int jobRef = ref_idx, jobQry = qry_idx;
        int mtHits = 0, mtInline = 0;
        bool suspended = false, mtSkip = false;
        unsigned long long mtCells0 = 0;
        if constexpr (MT == 3) {          // where the previous stitch launch left this pair
            const int32_t *fr = a.mt_front + (size_t)mtx * 8;
            const int fs = __builtin_amdgcn_readfirstlane(fr[0]);
            if (fs == 2) { mtSkip = true; last_tile = true; }
            else if (fs == 1) {
                tile = __builtin_amdgcn_readfirstlane(fr[1]); ref_idx = __builtin_amdgcn_readfirstlane(fr[2]); qry_idx = __builtin_amdgcn_readfirstlane(fr[3]);
                pos = __builtin_amdgcn_readfirstlane(fr[4]);
                mtCells0 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(fr[6]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(fr[7]) << 32);
            }
        }
        int8_t *out = (MT == 1) ? a.mt_seg + ((size_t)mtx * a.mt_slots + slot) * (size_t)a.mt_segcap : a.aln + (size_t)pair * 2 * (size_t)a.seq_len;
        unsigned long long cells = 0;
        long long steps_left = (long long)(R + Q + 2) * ((R + Q) / (max(marker, 2) - 1) + 4) + a.step_slack;
        // The row tags of the reductions hold k + 1 in 16 bits and a row in 16 bits -- TILE-local values, so sequences of any length pass
        // (a tile that has not converged after 65 534 diagonals goes to the round-1 kernel, below); the mailbox words of the speculative
        // start carry absolute positions in 16 bits each (the host does not pick that kernel for longer sequences).
        if (SP == 1 && !last_tile && !denomOne) { err = kErrGuard; last_tile = true; }      // (the leaf x leaf step has no division)
        if (!last_tile && ((SPEC && (R > 65535 || Q > 65535)) || !(denom >= 1.0f && denom <= 1.0995116e12f))) { err = (denom >= 1.0f && denom <= 1.0995116e12f) ? kErrOverflow : kErrGuard; last_tile = true; }
        int dbg_lastk = 0, dbg_conv = 0, dbg_L = 0, dbg_U = 0;
        bool guardBad = false;
        // SPEC: this workgroup runs the tiles of its parity; `confirmed` = the start of the tile in flight is the true one
        unsigned long long *team = SPEC ? a.team + (size_t)item * kTeamWords : nullptr;
        const int role = SPEC ? (int)(blockIdx.x & 1) : 0;
        bool confirmed = true, teamExit = false, redo = false, iEnded = false, timedOut = false;
        if constexpr (SPEC) tile = role;
        // broadcast of a 64-bit word from thread 0 to the workgroup
        auto bcast = [&](unsigned long long v) __attribute__((always_inline)) -> unsigned long long {
            if (threadIdx.x == 0) s_team[0] = v;
            __syncthreads();
            const unsigned long long r = s_team[0];
            __syncthreads();
            return (unsigned long long)__builtin_amdgcn_readfirstlane((int)(r & 0xFFFFFFFFu)) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(r >> 32)) << 32);
        };
        // SPEC: wait for the true start of tile `tile` and compare it with the start in use.  Returns 0 confirmed, 1 restart from
        // (ref_idx, qry_idx) as updated here, 2 the alignment is over (the partner finished or failed)
        auto settle = [&]() __attribute__((always_inline)) -> int {
            unsigned long long v = 0ull;
            if (threadIdx.x == 0) v = team_wait(&team[kTeamTruth + (tile & 3)], (unsigned)(tile + 1));
            v = bcast(v);
            if (v == 0ull) { err = kErrOverflow; timedOut = true; return 2; }
            const unsigned fl = (unsigned)(v >> 32) & 0xFFFFu;
            if (fl & (kTeamLast | kTeamErr)) return 2;
            const int tr = (int)((v >> 16) & 0xFFFFu), tq = (int)(v & 0xFFFFu);
            if (tr == ref_idx && tq == qry_idx) { if (threadIdx.x == 0) atomicAdd(&team[kTeamStat + 1], 1ull); return 0; }
            ref_idx = tr; qry_idx = tq;
            return 1;
        };
#ifdef TWL_KERNEL_STAMPS
        unsigned long long st_slots = 0, st_bar = 0, st_post = 0, st_n = 0, st_act = 0, st_exit = 0, st_setup = 0;
        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

        while (!last_tile) {   // ---- Align_freq tile loop, TALCO-XDrop.cpp:77-106 ----
#ifdef TWL_KERNEL_STAMPS
            const unsigned long long st_tile0 = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (SPEC) {
                if (!redo && tile > 0) {       // the start of this tile: the true one if it is known already, else the partner's guess
                    unsigned long long v = 0ull;
                    if (threadIdx.x == 0) v = team_wait(&team[kTeamTruth + (tile & 3)], (unsigned)(tile + 1), &team[kTeamGuess + (tile & 3)]);
                    v = bcast(v);
                    const unsigned fl = (unsigned)(v >> 32) & 0x7FFFu;
                    if (v == 0ull) { err = kErrOverflow; timedOut = true; teamExit = true; }
                    else if (!(v & (1ull << 47)) && (fl & (kTeamLast | kTeamErr))) teamExit = true;
                    else { ref_idx = (int)((v >> 16) & 0xFFFFu); qry_idx = (int)(v & 0xFFFFu); confirmed = !(v & (1ull << 47)); }
                }
                redo = false;
                if (teamExit) { last_tile = true; }
            }
            bool memoHit = false;
            if constexpr (MT == 3) {      // the tile that starts here may have been computed already (MT 1 launch): same start, same tile
                if (tile < a.mt_slots) {
                    const int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + tile) * kMtRec;
                    const int rcValid = __builtin_amdgcn_readfirstlane(rc[0]), rcRef = __builtin_amdgcn_readfirstlane(rc[1]), rcQry = __builtin_amdgcn_readfirstlane(rc[2]);
                    memoHit = (rcValid == 1 && rcRef == ref_idx && rcQry == qry_idx);
                    // ... or the tile job FAILED from this very start (its band outgrew the window, X-drop emptied it, ...): computing it in line
                    // again would end the same way some thousand diagonals later -- the pair takes the job's verdict
                    // (a band that outgrew the job's window says nothing when this launch has a wider one: protein tiles on 8 waves x 1 block)
                    if (rcValid == 2 && rcRef == ref_idx && rcQry == qry_idx && !(__builtin_amdgcn_readfirstlane(rc[10]) == kErrOverflow && __builtin_amdgcn_readfirstlane(rc[11]) < WINDOW)) {
                        err = __builtin_amdgcn_readfirstlane(rc[10]); if (err == 0) err = 3;
                        cells += (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rc[9]);      // (the band cells of the failed tile count, as they do in line)
                    }
                    if (memoHit) {
                        const int cnt = __builtin_amdgcn_readfirstlane(rc[6]), tailDir = __builtin_amdgcn_readfirstlane(rc[7]), tailLen = __builtin_amdgcn_readfirstlane(rc[8]);
                        if (pos + cnt + tailLen > 2 * a.seq_len) err = 3;
                        else {
                            const int8_t *sg = a.mt_seg + ((size_t)mtx * a.mt_slots + tile) * (size_t)a.mt_segcap;
                            for (int t = threadIdx.x; t < cnt; t += C::THREADS) out[pos + t] = sg[t];
                            for (int t = threadIdx.x; t < tailLen; t += C::THREADS) out[pos + cnt + t] = (int8_t)tailDir;
                            pos += cnt + tailLen;
                            ref_idx = __builtin_amdgcn_readfirstlane(rc[3]); qry_idx = __builtin_amdgcn_readfirstlane(rc[4]);
                            last_tile = __builtin_amdgcn_readfirstlane(rc[5]) != 0;
                            cells += (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(rc[9]);
                            mtHits += 1;
                            tile += 1;
                        }
                    }
                }
                if (err != 0) break;                                            // (a failed record: the pair ends here, in whichever round)
                if (!memoHit && !a.mt_inline) { suspended = true; break; }      // not the last round: the next one predicts again from here
                if (!memoHit) mtInline += 1;
            }
            if (!(SPEC && teamExit) && !memoHit) {
            const int refLen = R - ref_idx, qLen = Q - qry_idx;
            const int fLen = min(a.flen, min(refLen, qLen));                          // :258
            // one unsigned compare per step covers "band empty", "wider than fLen" and (conservatively) "outgrew the window"
            // (two blocks of margin: below it a block that leaves the band cannot be needed again on the same diagonal)
            const int fcap = min(fLen, 64 * (NV - 2));
            // the nucleotide matrix values of mode 2 (match / transition / transversion) in vector registers: see the column score of the geometries that form the first products per cell
            float vmA = a.M[0], vmB = a.M[2], vmC = a.M[1];
            asm volatile("" : "+v"(vmA), "+v"(vmB), "+v"(vmC));
            // ---- per-slot state ----
            float S1[RPL], I1[RPL], D1[RPL], LS2[RPL];
            int CS1[RPL], CI1[RPL], CD1[RPL], LCS2[RPL];
            float qv[RPL][QN], gopq[RPL], gexq[RPL], qM[RPL][NQM];
            nuc_f2 qP[QPRE ? RPL : 1][NQP];           // qP[2*m + h] = {q[m]*M[2h][m], q[m]*M[2h+1][m]}
            float simNext[RPL];            // presim: the score of this row on the NEXT diagonal, loaded one diagonal ahead
            int simFor[RPL];               // ... and the diagonal it belongs to
            const int simK0 = ref_idx + qry_idx;   // global anti-diagonal of the tile's first cell
            int b0[RPL];                   // first row of the 64-row block this slot holds (a scalar: block index * 64)
            unsigned ra[RPL];              // byte address of this lane's reference column in plane 0 of the ring
            uint32_t tbacc[RPL];
            bool q5any[RPL];

            auto load_q = [&](int r) __attribute__((always_inline)) {
                const int i = b0[r] + lane;
                const bool ok = qry_idx + i < Q;
                if constexpr (PRESIM) {
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) c = colsQ[F4 * (size_t)(qry_idx + i) + (F4 - 1)];
                    gopq[r] = c.z; gexq[r] = c.w;
                    q5any[r] = false;
                    simFor[r] = -1;
                } else {
                    float cb[4 * F4];
#pragma unroll
                    for (int t = 0; t < F4; ++t) {
                        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) c = colsQ[F4 * (size_t)(qry_idx + i) + t];
                        cb[4 * t] = c.x; cb[4 * t + 1] = c.y; cb[4 * t + 2] = c.z; cb[4 * t + 3] = c.w;
                    }
#pragma unroll
                    for (int t = 0; t < P; ++t) qv[r][t] = cb[t];
                    gopq[r] = cb[P]; gexq[r] = cb[P + 1];
                    // first rounding of (q[m]*M[l][m])*r[l], :386
                    if constexpr (MM == 2 && !QPRE) {}
                    else if constexpr (MM == 2) {
                        const float mA = a.M[0], mB = a.M[2], mC = a.M[1];
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const int l0 = 2 * h, l1 = 2 * h + 1;
                                qP[r][2 * m + h] = nuc_f2{qv[r][m] * ((l0 == m) ? mA : (((l0 ^ m) == 2) ? mB : mC)), qv[r][m] * ((l1 == m) ? mA : (((l1 ^ m) == 2) ? mB : mC))};
                            }
                    } else if constexpr (MM == 1) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int h = 0; h < 2; ++h) qP[r][2 * m + h] = nuc_f2{qv[r][m] * a.M[5 * (2 * h) + m], qv[r][m] * a.M[5 * (2 * h + 1) + m]};
                    } else if constexpr (MM == 5) {
                        // Query rows with ONE non-zero letter (a single sequence; the host says so): of the four products of a matrix row only
                        // that letter's is not +-0, so s_l = ((t0 + t1) + t2) + t3 collapses to (q[m*] * M[l][m*]) * r[l] exactly.  Kept per row:
                        // qT[l] = the row sum of the first products (= q[m*] * M[l][m*]); the score is then four products and three sums.
                        float qT[4];
#pragma unroll
                        for (int l = 0; l < 4; ++l) qT[l] = ((qv[r][0] * a.M[5 * l + 0] + qv[r][1] * a.M[5 * l + 1]) + qv[r][2] * a.M[5 * l + 2]) + qv[r][3] * a.M[5 * l + 3];
                        qP[r][0] = nuc_f2{qT[0], qT[1]};
                        qP[r][1] = nuc_f2{qT[2], qT[3]};
                    } else if constexpr (MM == 0) {
#pragma unroll
                        for (int m = 0; m < 5; ++m) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) qP[r][2 * m + h] = nuc_f2{qv[r][m] * a.M[5 * (2 * h) + m], qv[r][m] * a.M[5 * (2 * h + 1) + m]};
                            qM[r][m] = qv[r][m] * a.M[20 + m];
                        }
                    }
                    if constexpr (SP == 1) q5any[r] = false;
                    else q5any[r] = gcNZ && __builtin_amdgcn_ballot_w64(cb[P - 1] != 0.0f) != 0ull;
                    bool bad = false;
                    if constexpr (SP == 1) bad = (cb[P - 1] != 0.0f);      // a gap letter in a "leaf" row: the promise does not hold
#pragma unroll
                    for (int t = 0; t < P; ++t) bad = bad | div_guard_bad(cb[t]);
                    guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
                }
            };
            auto load_ring_block = [&](int B) __attribute__((always_inline)) {
                const int col = 64 * B + lane;
                const int slot = (B % NB) * 64 + lane;
                const bool ok = col < refLen;
                if constexpr (PRESIM) {
                    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) c = colsR[F4 * (size_t)(ref_idx + col) + (F4 - 1)];
                    s_ring[slot] = c;
                } else {
                    bool bad = false;
                    uint32_t mk = 0;
#pragma unroll
                    for (int t = 0; t < F4; ++t) {
                        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) c = colsR[F4 * (size_t)(ref_idx + col) + t];
                        s_ring[t * CAP + slot] = c;
                        const float f[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (4 * t + u < P) bad = bad | div_guard_bad(f[u]);
                            if (SP == 1 && 4 * t + u == P - 1) bad = bad | (f[u] != 0.0f);
                            if (SPARSE && 4 * t + u < 21) mk |= (f[u] != 0.0f) ? (1u << (4 * t + u)) : 0u;
                        }
                    }
                    if constexpr (SPARSE) s_rmask[slot] = mk;
                    guardBad = guardBad | (__builtin_amdgcn_ballot_w64(bad) != 0ull);
                }
            };
            auto ring_addr = [&](int r, int k) __attribute__((always_inline)) {      // ((k - 64*blk - lane) mod CAP) * 16
                int rs = (k - b0[r] - lane) % CAP;
                rs += (rs < 0) ? CAP : 0;
                ra[r] = (unsigned)rs * 16u + lds_off(s_ring);
            };

#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                b0[r] = 64 * (r * W + w);
                ring_addr(r, 0);
                tbacc[r] = 0;
                S1[r] = I1[r] = D1[r] = LS2[r] = -1.0f;       // never read before written for in-band cells
                load_q(r);
            }
            int hiBlk = 1;
            if (w == 0 % W) load_ring_block(0);
            if (w == 1 % W) load_ring_block(1);
            for (int t = threadIdx.x; t < WINDOW + 4; t += C::THREADS) { s_par[0].cd[t] = kDB; s_par[1].cd[t] = kDB; }      // :308
            if (threadIdx.x == 0) {
                s_par[0].red[0] = s_par[1].red[0] = (unsigned)__float_as_int(-inf);
                s_par[0].red[1] = s_par[0].red[2] = s_par[1].red[1] = s_par[1].red[2] = 0u;
                for (int t = 0; t < 2; ++t) { s_par[t].conv[0] = 0x7fffffff; s_par[t].conv[1] = (int)0x80000000; s_par[t].conv[2] = 0; }
                s_team[1] = 0ull; s_team[2] = 0ull; s_team[3] = 0ull;
                s_misc[4] = 0;
            }
            __syncthreads();

            // ---- Tile, TALCO-XDrop.cpp:233-689 ----
            // The band bookkeeping lives twice: as "uniform vectors" (v*: the same value in every lane, in vector registers, so
            // that the per-diagonal arithmetic and every lane test run on the four vector units) and, for the few branches, as the
            // scalars Lk / Uk read back once per diagonal.  Bands: diagonal k as (vL, vU); diagonal k-1 as (vlo1, vw1 = width-1),
            // diagonal k-2 as (vlo2p = low+1, vw2); an empty band is (0x3fffffff, 0), which no row matches (:296-297 start with
            // L > U for k-1 and k-2).
            int vL = 0, vU = 0, vlo1 = 0x3fffffff, vlo2p = 0x3fffffff;
            unsigned vw1 = 0, vw2 = 0;
            int vwid1 = 0;                               // width of diagonal k-1 (0 when empty): the stale CD slot, see s_cd
            unsigned vcells = 0;                         // < 2^32 per tile
            // Scalars of the band of diagonal k, read back once per diagonal: its low end, and what the activity test of a slot compares against --
            // Lm63 = Lk - 63, W64 = Uk - Lk + 64 (a block takes part when 0 <= b - Lm63 <= W64; it has fallen out of the band when b - Lm63 < 0).
            // Round 6: the high end itself is only formed where it is rare (hook, stop decoding, first row / column).
            int Lk = 0, Wk = 0, Lm63 = -63, W64 = 64;
            int kStop = 0;                               // the innermost loops run while k < kStop; whatever ends the tile also pulls it down
            int kHook = 8;                               // the next k at which the every-8th-diagonal hook runs (a compare of two scalars per diagonal)
            float msp = -inf, convf = 0.0f;              // running maximum (:259), score at convergence (:594)
            bool converged = false;
            int conv_value = 0, prev_conv_s = -1;
            const int kEnd = refLen + qLen - 1;
            int k = 0;
            int tile_err = 0;
            bool go = true, conv_logic = false, aborted = false;
            bool spec = true;                            // the next diagonal may be a "special" one (k == 0, or tile 0's first row/column)
            bool tbPending = false;
            unsigned tbOff = (unsigned)lane * 4u;        // byte offset of this lane's word in the current group of 8 diagonals (slot 0)
            // Round 5: the traceback word of a slot is stored only when its block can have held band cells in the group of 8 diagonals (until then every window row
            // wrote its word every 8th diagonal: 69 % of the path's HBM traffic).  L never decreases and U grows by at most a row per diagonal, so the bands of a
            // group lie within [L, U + 8] of the diagonal in front of it (tbL0, tbU0); a slot that moved to another block in the group stores anyway (tbMust).
            // The walk only reads words of cells on the path, and those are band cells.
            int tbL0 = 0, tbU0 = 0;
            unsigned tbMust = 0u;
            // Parity: vcur is the struct of diagonal k, vprev that of k-1; they swap by one xor each per diagonal.
            constexpr unsigned PARX = (unsigned)sizeof(ParBuf);      // s_par[1] - s_par[0]
            unsigned vcur = lds_off(&s_par[0]), vprev = lds_off(&s_par[1]);
            const unsigned parx = lds_off(&s_par[0]) ^ lds_off(&s_par[1]);
            (void)PARX;
            // Single-lane LDS side effects without touching EXEC: every lane issues the instruction, the lanes that have nothing
            // to say aim at their trash slot of the same struct.
            // (+ O_RED, the immediate of the reduction atomics, = a trash word of this lane; consecutive words: 64 lanes over 32 banks,
            // where the 16-byte slots of the mailbox stores made every atomic an 8-way bank conflict)
            const unsigned relTrashRed = O_TRASH + (unsigned)lane * 4u - O_RED;
            unsigned mbRel[RPL], exRel[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int vw = r * W + w;
                mbRel[r] = (lane == 63) ? O_EXCH + 16u * (unsigned)vw : O_TRASH + (unsigned)lane * 16u;
                exRel[r] = O_EXCH + 16u * (unsigned)((vw + NV - 1) % NV);
            }

            // One diagonal.  PH: 0 = A (k < marker-1), 1 = B (k == marker-1 or marker), 2 = C (k > marker).
            // Ends the loops by clearing `go`: tile_err on a stop condition, conv_logic when the tile ended by convergence (:609-612).
            // GEN (round 6): true = the general step, which tests for the special diagonals (k == 0, tile 0's first row / column); false = the plain step the
            // loops below switch to once `spec` has gone false for good: the test does not exist in it.  The every-8th-diagonal hook runs at the END of a step,
            // behind the band update (until round 5 in front of the barrier, behind a five-instruction test).
            auto stop_tile = [&]() __attribute__((always_inline)) { go = false; kStop = (int)0x80000000; };
            auto hook = [&](auto PHtag) __attribute__((always_inline)) {
                constexpr int PH = decltype(PHtag)::value;
                constexpr bool TB = (PH != 2);
                const int Uk = Lk + Wk;
                if constexpr (TB) {
#pragma unroll
                    for (int r = 0; r < RPL; ++r) {
                        const int bb = b0[r];
#if defined(TWL_TB_ALWAYS)      // experiment builds: every window row stores its word, as until round 4
                        (void)bb;
#else
                        if (((tbMust >> r) & 1u) != 0u || (bb + 63 >= tbL0 && bb <= tbU0 + 8))
#endif
                            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (r * W + w))) = tbacc[r];
                        tbacc[r] = 0;
                    }
                    tbMust = 0u; tbL0 = Lk; tbU0 = Uk;
                    tbOff += (unsigned)WINDOW * 4u;
                    tbPending = false;
                }
                // every 8th diagonal: stage the next 64 reference columns when the band gets within 10 + 64 columns of them (the ring has
                // two blocks more than the window, so a block loaded this early never overwrites one still in use).  The hook runs behind the band update of
                // diagonal k - 1, k already counted up: the columns are in LDS at the barrier of diagonal k, a diagonal before the next hook's range begins
                const int need_hi = ((k + 10 - Lk) >> 6) + 1;
                if (hiBlk < need_hi) { ++hiBlk; if (w == hiBlk % W) load_ring_block(hiBlk); }
            };
            auto step = [&](auto PHtag, auto GENtag) __attribute__((always_inline)) {
                constexpr int PH = decltype(PHtag)::value;
                constexpr bool GEN = decltype(GENtag)::value;
                constexpr bool TB = (PH != 2), CONV = (PH != 0);
                asm volatile("; TWL_STEP PH=%c0 GEN=%c1 (a comment: tools/isa_block_step.py finds the diagonal loops by it)" ::"n"(PH), "n"((int)GEN));
                TWL_STAMP(t_head);
#if defined(TWL_EXP_SALU)      // timing experiment: 24 extra scalar instructions per wave and diagonal
                { int xs = k; 
#pragma unroll
                  for (int t = 0; t < 24; ++t) asm volatile("s_add_u32 %0, %0, 1" : "+s"(xs));
                  asm volatile("" :: "s"(xs)); }
#elif defined(TWL_EXP_VALU)    // timing experiment: 24 extra vector instructions per wave and diagonal
                { int xv = lane;
#pragma unroll
                  for (int t = 0; t < 24; ++t) asm volatile("v_add_u32 %0, %0, 1" : "+v"(xv));
                  asm volatile("" :: "v"(xv)); }
#endif
                const unsigned kk16 = (unsigned)(k + 1) << 16;                // tag of this diagonal's row reductions (k + 1 <= 65535)
                const unsigned vwidth1 = (unsigned)(vU - vL);
                vcells += vwidth1 + 1u;
                // :495 with :607 (max(x, 0) of a float is max of its bits as an integer).  The X-drop as a SCALAR operand (v_subrev: vsrc1 - src0): left to the
                // compiler it sat in a vector register that the 96-register kernels reloaded from scratch at the head of every diagonal
                float thr;
                { const float m0 = __int_as_float(max(__float_as_int(msp), 0)); asm("v_subrev_f32_e32 %0, %1, %2" : "=v"(thr) : "s"(xdropf), "v"(m0)); }
                bool special = false;
                if constexpr (GEN) {
                    if (__builtin_expect(spec, 0)) {
                        special = (k == 0) | ((tile == 0) && (Lk == 0 || Lk + Wk == k));
                        spec = special;                                      // both conditions are monotone: once false, false for the tile
                    }
                }
                int staleCD = kDB;
                if constexpr (PH == 2) staleCD = lds_ld<int>(vprev + 4u * (unsigned)vwid1 + O_CD);   // one broadcast read per diagonal
                const unsigned vTrashRed = vcur + relTrashRed;

#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int b = b0[r];
                    const int u = b - Lm63;
                    // the block takes part when the band touches it or will reach it on the next diagonal (its lane 0 then needs
                    // S[k-1][b-1] now, to have S[k-2][i-1] next time): Lk - 63 <= b <= Uk + 1
                    if ((unsigned)u <= (unsigned)W64) {
                        // a wave with cells to compute issues ahead of the idle waves of its SIMD (their per-diagonal bookkeeping otherwise sits
                        // in front of it: an active wave started its step up to 1100 cycles after the first wave of its SIMD); back to 0
                        // in front of the barrier, where the idle waves must not be held up -- keeping the priority across the barrier
                        // gave the gain away again.  Wide level 130 -> 121 ms, 100 pairs in tiles 8.2 -> 7.6 ms (tools/exp_step_cost.py).
                        // ... and a wave that enters a SECOND block of the same diagonal is the one its workgroup will wait for: top priority from there on
                        // (round 4, throughput launches: 2048 pairs of 10 kbp 95.8 -> 93.0 ms; in the tile jobs of the tile-parallel path the same cost 8 %: not there)
                        // (round 6: by slot index, not by a count of the active slots -- the count cost up to eight scalar instructions per diagonal; a wave whose
                        //  only block sits in a later slot takes the top priority too, which it gives back at the barrier)
                        if constexpr (MT == 0 && RPL >= 2) { if (r == 0) TWL_SETPRIO(2); else TWL_SETPRIO(3); }
                        else TWL_SETPRIO(2);
                        const int i = b + lane;
                        // ---- loads: mailbox of the previous block, reference column of this cell ----
                        float eS, eI; int eCS = 0, eCI = 0;
                        if constexpr (CONV) {
                            const nuc_i4 e = lds_ld<nuc_i4>(vprev + exRel[r]);
                            eS = __int_as_float(e.x); eI = __int_as_float(e.y); eCS = e.z; eCI = e.w;
                        } else {
                            const nuc_i2 e = lds_ld<nuc_i2>(vprev + exRel[r]);
                            eS = __int_as_float(e.x); eI = __int_as_float(e.y);
                        }
                        const bool inband = (unsigned)(i - vL) <= vwidth1;
                        float numer = 0.0f, rg, gopr, gexr;
                        if constexpr (P == 6) {
                            const float4 c0 = lds_ld128(ra[r]);
                            const float4 c1 = lds_ld128(ra[r] + CAP * 16);
                            const float rc[5] = {c0.x, c0.y, c0.z, c0.w, c1.x};
                            rg = c1.y; gopr = c1.z; gexr = c1.w;
                            // ---- column score, :378-395 (order: per l the products summed left to right, accumulated over l) ----
                            // rows {0,1} and {2,3} side by side: t[l][m] = (q[m]*M[l][m]) * r[l]; s_l = ((t0 + t1) + t2) + t3 (+ t4)
                            constexpr int NM = (MM == 0) ? 5 : (MM == 5 ? 1 : 4);
                            const nuc_f2 r01 = nuc_f2{c0.x, c0.y}, r23 = nuc_f2{c0.z, c0.w};
                            nuc_f2 s01, s23;
                            if constexpr (QPRE) {
                                s01 = qP[r][0] * r01; s23 = qP[r][1] * r23;
#pragma unroll
                                for (int m = 1; m < NM; ++m) { s01 = s01 + qP[r][2 * m] * r01; s23 = s23 + qP[r][2 * m + 1] * r23; }
#if defined(TWL_EXP_PACKED_SCORE)      // experiment builds: the packed form of rounds 4-5 (matrix values as scalar operands, v_pk_mul / v_pk_add)
                            } else if constexpr (true) {
                                const float mA = a.M[0], mB = a.M[2], mC = a.M[1];
                                auto fp = [&](int m, int h) __attribute__((always_inline)) {
                                    const int l0 = 2 * h, l1 = 2 * h + 1;
                                    return nuc_f2{qv[r][m] * ((l0 == m) ? mA : (((l0 ^ m) == 2) ? mB : mC)), qv[r][m] * ((l1 == m) ? mA : (((l1 ^ m) == 2) ? mB : mC))};
                                };
                                s01 = fp(0, 0) * r01; s23 = fp(0, 1) * r23;
#pragma unroll
                                for (int m = 1; m < 4; ++m) { s01 = s01 + fp(m, 0) * r01; s23 = s23 + fp(m, 1) * r23; }
#endif
                            } else {      // the first products of this row formed here (see QPRE): same operations, same order
                                // Round 6: as plain fp32 operations on VECTOR registers.  Measured (tools/micro/issue_rates5.hip, profiles/r06/issue_rates5.log): an fp32
                                // add / mul / fma on vector registers issues at 3.6-3.8 per ns and CU, the same multiply with a scalar register as an operand at 2.2 (as
                                // does every integer, compare, select and DPP form), a packed one at 1.85.  The three matrix values live in vector registers
                                // (vmA / vmB / vmC), the twelve distinct first products q[m] * M and the sixteen second products are plain multiplies: 43 full-rate
                                // instructions where sixteen scalar-operand multiplies and fourteen packed ones stood.  Same operations in the same order;
                                // 2048 pairs of 10 kbp profiles 91.2 -> 86.5 ms (profiles/r06/exp_step_variants.txt)
                                auto mv = [&](int l, int m) __attribute__((always_inline)) { return (l == m) ? vmA : (((l ^ m) == 2) ? vmB : vmC); };
                                float sl[4];
#pragma unroll
                                for (int l = 0; l < 4; ++l) {
                                    sl[l] = (qv[r][0] * mv(l, 0)) * rc[l];
#pragma unroll
                                    for (int m = 1; m < 4; ++m) sl[l] = sl[l] + (qv[r][m] * mv(l, m)) * rc[l];
                                }
                                s01 = nuc_f2{sl[0], sl[1]}; s23 = nuc_f2{sl[2], sl[3]};
                            }
                            numer = ((s01.x + s01.y) + s23.x) + s23.y;
                            if constexpr (MM == 0) {
                                float s4 = qM[r][0] * rc[4];
#pragma unroll
                                for (int m = 1; m < 5; ++m) s4 = s4 + qM[r][m] * rc[4];
                                numer = numer + s4;
                            }
                            // the gap-letter terms, :394-395.  With gapCharScore 0 (the deferred pass, groups of > 10 000 sequences: alignment-cpu.cpp:88)
                            // every one of them is +-0 and the running sum keeps its value (up to the sign of a zero, which nothing downstream sees)
                            // (the test of gapCharScore sits in q5any and in the mask below: as a branch of its own around the two blocks it cost the wide level 2.5 %)
                            if (q5any[r]) {
#pragma unroll
                                for (int l = 0; l < 5; ++l) numer += (rc[l] * qv[r][5]) * gc;          // :394
                            }
                            bool rgAny;
                            if constexpr (SP == 1) rgAny = false;
                            else rgAny = (__builtin_amdgcn_ballot_w64(inband) & __builtin_amdgcn_ballot_w64(rg != 0.0f) & gcMask) != 0ull;
                            if (rgAny) {
#pragma unroll
                                for (int m = 0; m < 5; ++m) numer += (rg * qv[r][m]) * gc;             // :395
                            }
                        } else if constexpr (SPARSE) {
                            // protein column score, :409-433, over the NON-ZERO letters of the reference column only: a skipped letter has r[l] == 0,
                            // so each of its products and their block sum are +-0, and adding +-0 to a running sum that started at +0 changes
                            // nothing.  Letters in ascending order (the reference's order); per letter the tail m = 16..20 first, then the two
                            // 8-lane blocks v[t] = (q[t]*M[l][t])*r[l] + (q[8+t]*M[l][8+t])*r[l] summed left to right.
                            const float4 cl = lds_ld128(ra[r] + (F4 - 1) * CAP * 16);                    // {X letter, gap, gapOpen, gapExtend}
                            rg = cl.y; gopr = cl.z; gexr = cl.w;
                            const unsigned slotOff = ra[r] - lds_off(s_ring);                             // 16 * ring slot
                            const uint32_t rmask = inband ? lds_ld<uint32_t>(lds_off(s_rmask) + (slotOff >> 2)) : 0u;
                            uint32_t mk = rmask;
                            while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                                if (mk != 0u) {
                                    const int l = __builtin_ctz(mk);
                                    mk &= mk - 1u;
                                    const float rl = lds_ld<float>(ra[r] + (unsigned)(l >> 2) * (CAP * 16) + (unsigned)(l & 3) * 4u);
                                    float Mr[24];
#pragma unroll
                                    for (int t = 0; t < 6; ++t) {
                                        const float4 c = s_M4[l * 6 + t];
                                        Mr[4 * t] = c.x; Mr[4 * t + 1] = c.y; Mr[4 * t + 2] = c.z; Mr[4 * t + 3] = c.w;
                                    }
#pragma unroll
                                    for (int m = 16; m < 21; ++m) numer += (rl * qv[r][m]) * Mr[m];
                                    float v[8];
#pragma unroll
                                    for (int t = 0; t < 8; ++t) v[t] = (qv[r][t] * Mr[t]) * rl + (qv[r][8 + t] * Mr[8 + t]) * rl;
                                    numer += ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
                                }
                            }
                            if (q5any[r]) {              // (r[l]*q[gap])*gc is +-0 for the skipped letters, :432
                                mk = rmask;
                                while (__builtin_amdgcn_ballot_w64(mk != 0u) != 0ull) {
                                    if (mk != 0u) {
                                        const int l = __builtin_ctz(mk);
                                        mk &= mk - 1u;
                                        numer += (lds_ld<float>(ra[r] + (unsigned)(l >> 2) * (CAP * 16) + (unsigned)(l & 3) * 4u) * qv[r][P - 1]) * gc;
                                    }
                                }
                            }
                            if ((__builtin_amdgcn_ballot_w64(inband) & __builtin_amdgcn_ballot_w64(rg != 0.0f)) != 0ull) {
#pragma unroll
                                for (int m = 0; m < P - 1; ++m) numer += (rg * qv[r][m]) * gc;        // :433
                            }
                        } else {
                            const float4 cl = lds_ld128(ra[r]);                                          // {X letter, gap, gapOpen, gapExtend}
                            rg = cl.y; gopr = cl.z; gexr = cl.w;
                            (void)rg;
                        }
                        float sim = numer;                                                         // :444
                        if constexpr (PRESIM) {
                            // the score was computed ahead by score_matrix_kernel (same arithmetic); this row's value for THIS diagonal was requested
                            // one diagonal ago, the one for the next diagonal is requested now and arrives while the workgroup synchronises
                            const size_t col = (size_t)(qry_idx + i);
                            const bool colOk = (qry_idx + i) < simPitch;
                            if (simFor[r] == k) sim = simNext[r];
                            else sim = colOk ? simP[(size_t)(k + simK0) * (size_t)simPitch + col] : 0.0f;
                            simNext[r] = (colOk && k + 1 < kEnd) ? simP[(size_t)(k + 1 + simK0) * (size_t)simPitch + col] : 0.0f;
                            simFor[r] = k + 1;
                            // Round 6: score_matrix_kernel fills a corridor around the straight line between the corners and leaves NaN in the tiles outside it.  A band
                            // cell that reads one sends the pair to the kernel that scores in line (kErrGuard: the stage an operand outside fast_div's range takes)
                            guardBad = guardBad | (__builtin_amdgcn_ballot_w64(inband && sim != sim) != 0ull);
                        } else if (SP == 0 && !denomOne) {      // (a real branch: leaf pairs, a third of all cells, have refNum * qryNum == 1)
                            sim = fast_div(numer, denom, rden);
                            asm volatile("" : "+v"(sim));
                        }

                        if constexpr (DUMP) {      // diagnostics build of the same code: the score of every cell the band visits
                            if (inband) a.simdump[(size_t)(qry_idx + i) * (size_t)R + (size_t)(ref_idx + k - i)] = sim;
                        }
                        // ---- neighbours ----
                        const float LS1 = dpp_shr1_f(eS, S1[r]);
                        const float LI1 = dpp_shr1_f(eI, I1[r]);
                        int LCS1 = 0, LCI1 = 0;
                        if constexpr (CONV) { LCS1 = dpp_shr1_i(eCS, CS1[r]); LCI1 = dpp_shr1_i(eCI, CI1[r]); }
                        const unsigned t1 = (unsigned)(i - vlo1);
                        const bool up_ok = t1 <= vw1;                      // i   in band(k-1)
                        const bool left_ok = (t1 - 1u) <= vw1;             // i-1 in band(k-1)
                        const bool diag_ok = (unsigned)(i - vlo2p) <= vw2; // i-1 in band(k-2)
                        // ---- recurrence, :445-497 ----
                        float match = diag_ok ? LS2[r] + sim : -inf;
                        if (__builtin_expect(special, 0)) {
                            const int j = k - i;
                            if (k == 0) match = sim;
                            else if (i == 0 || j == 0) {
                                int far = max(i, j) - 1; far = far < 0 ? 0 : far;
                                match = (sim + a.gap_open) + a.gap_extend * (float)far;
                            }
                        }
                        const float delOp = S1[r] + gopr, delExt = D1[r] + gexr;                  // :456-463
                        const float insOp = LS1 + gopq[r], insExt = LI1 + gexq[r];
                        // an invalid neighbour makes both candidates -inf: the state is -inf and "extend" wins the tie (:468-475).
                        // (compare + select rather than fmaxf: the compare is needed for the pointer bits anyway, and a max would be
                        // preceded by two canonicalising moves)
                        const bool dGE = delExt >= delOp, iGE = insExt >= insOp;
                        const bool Dptr = !up_ok | dGE;
                        const bool Iptr = !left_ok | iGE;
                        const float Dv = up_ok ? (dGE ? delExt : delOp) : -inf;
                        const float Iv = left_ok ? (iGE ? insExt : insOp) : -inf;
                        // :477-494: G = (I > D ? I : D) is the better gap state (D wins ties); M wins when match >= G
                        const bool gapIsI = Iv > Dv;
                        const float Gv = gapIsI ? Iv : Dv;
                        const bool isM = match >= Gv;
                        float Sv = isM ? match : Gv;
                        Sv = (Sv < thr) ? -inf : Sv;                                               // :495-497

                        if constexpr (CONV) {                                                      // :520-547
                            int CSn, CIn, CDn;
                            const int i16 = i & 0xFFFF;
                            if (PH == 1 && k == marker - 1) { CSn = (3 << 16) | i16; CIn = CI1[r]; CDn = CD1[r]; }
                            else if (PH == 1) { CSn = i16; CIn = (1 << 16) | i16; CDn = (2 << 16) | i16; }
                            else {
                                const int viaS = (LCS1 != -1) ? LCS1 : kIB;
                                CIn = left_ok ? (Iptr ? LCI1 : viaS) : kIB;
                                const int cdUp = up_ok ? CD1[r] : staleCD;      // above the stored band the reference reads the stale slot (:535)
                                CDn = Dptr ? cdUp : ((CS1[r] != -1) ? CS1[r] : kDB);
                                const int viaGap = gapIsI ? CIn : CDn;
                                CSn = isM ? (diag_ok ? LCS2[r] : -1) : viaGap;   // M without a diagonal predecessor: "unset", as in the oracle (:541)
                            }
                            CS1[r] = CSn; CI1[r] = CIn; CD1[r] = CDn;
                            if (PH == 2 || k == marker) { if (inband) lds_st<int>(vcur + 4u * (unsigned)(i - vL) + O_CD, CDn); }
                        }
                        S1[r] = Sv; I1[r] = Iv; D1[r] = Dv;
                        if constexpr (GUESS && PH == 1) {      // the guess for the next tile's start: best cell of diagonal marker-1 / marker
                            if (inband && Sv > -inf) {
                                const unsigned long long key = ((unsigned long long)score_key(Sv) << 32) | (unsigned)i;
                                asm volatile("ds_max_u64 %0, %1" ::"v"(lds_off(&s_team[(k == marker) ? 3 : 2])), "v"(key) : "memory");
                            }
                        }
                        // ---- reductions of this diagonal (:501-503, :563-583), all lanes issuing, see vTrash ----
                        const float Sin = inband ? Sv : -inf;             // out-of-band lanes take no part
                        lds_max_f32_off<O_RED>((Sin > msp) ? vcur : vTrashRed, Sin);
                        const unsigned long long vm = __builtin_amdgcn_ballot_w64(Sin > -inf);
                        // the first unpruned lane posts the low end, the last one the high end
                        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u));
                        const int rank = (Sin > -inf) ? below : -2;
                        const int nValid1 = (int)__builtin_popcountll(vm) - 1;
                        lds_max_u32_off<O_RED + 4>((rank == 0) ? vcur : vTrashRed, kk16 + (0xFFFFu - (unsigned)i));
                        lds_max_u32_off<O_RED + 8>((rank == nValid1) ? vcur : vTrashRed, kk16 + (unsigned)i);
                        if constexpr (CONV) lds_st<nuc_i4>(vcur + mbRel[r], nuc_i4{__float_as_int(Sv), __float_as_int(Iv), CS1[r], CI1[r]});
                        else lds_st<nuc_i2>(vcur + mbRel[r], nuc_i2{__float_as_int(Sv), __float_as_int(Iv)});
                        if constexpr (PH == 2) {       // CS of this block's first / last unpruned row, for the pre-test of the convergence test (below)
                            const unsigned trashE = vcur + O_TRASH + (unsigned)lane * 16u + 8u;
                            lds_st<int>((rank == 0) ? vcur + O_EDGE + 8u * (unsigned)(r * W + w) : trashE, CS1[r]);
                            lds_st<int>((rank == nValid1) ? vcur + O_EDGE + 8u * (unsigned)(r * W + w) + 4u : trashE + 4u, CS1[r]);
                        }
                        if constexpr (TB) {                                                        // :548-557
                            const uint32_t nib = (isM ? 0u : (gapIsI ? 1u : 2u)) | (Iptr ? 4u : 0u) | (Dptr ? 8u : 0u);
                            tbacc[r] |= nib << (4 * (k & 7));
                        }
                        LS2[r] = LS1;
                        if constexpr (CONV) LCS2[r] = LCS1;
                    }
                    if (__builtin_expect(u < 0, 0)) {    // block fell out of the band (b + 63 < Lk): take the next one
                        while (b0[r] + 63 < Lk) b0[r] += 64 * NV;
                        ring_addr(r, k);
                        load_q(r);
                        tbMust |= 1u << r;
                    }
                    ra[r] += 16u;
                    if (ra[r] == lds_off(s_ring) + CAP * 16u) ra[r] = lds_off(s_ring);
                }
                if constexpr (TB) tbPending = true;
                TWL_STAMP(t_slots);
                TWL_SETPRIO(0);
                wg_barrier_lds();
                if constexpr (W == 16) {   // one workgroup per CU (latency geometry): a wave that had cells on this diagonal also runs the band bookkeeping behind
                                           // the barrier ahead of the idle ones (lone pair 1.72 -> 1.67 ms); with two workgroups per CU the same cost 6 %
                    bool anyBlk = false;
#pragma unroll
                    for (int r = 0; r < RPL; ++r) anyBlk |= ((unsigned)(b0[r] - Lm63) <= (unsigned)W64);
                    if (anyBlk) TWL_SETPRIO(2);
                }
                TWL_STAMP(t_bar);

                // ---- post: the band of the next diagonal, :563-604 ----
                // {running max, low-end tag, high-end tag}: a tag from an older diagonal is smaller than this diagonal's, so when no
                // row survived the differences below come out as newU <= -1 and newL >= 65536: an empty band, caught by the next test
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
                                // (made inside the branch: hoisted out of the loop these four registers were spilled to scratch in the
                                // 128-register kernels and re-loaded by wave 0 on every diagonal of phase C)
                                int c0 = 0x7fffffff, c1 = (int)0x80000000, c2 = 0;
                                asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2));
                                lds_st<nuc_i4>(vprev + O_CONV, nuc_i4{c0, c1, c2, c2});
                            }
                            // Pre-test (a necessary condition, so the result is unchanged): the surviving band [newL, newU] can only be uniform
                            // when its two end cells hold the same convergence pointer.  The ends' pointers were left in LDS by their blocks
                            // before the barrier; two broadcast reads decide.  Until shortly before the tile converges the ends disagree, and
                            // the test proper -- ballots, three reductions, a second workgroup barrier -- is skipped: conv_S = -1, as it would find.
                            const int cLo = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newL >> 6) % (unsigned)NV));      // (a mask when NV is a power of two)
                            const int cHi = lds_ld<int>(vcur + O_EDGE + 8u * (((unsigned)newU >> 6) % (unsigned)NV) + 4u);
                            const bool maybe = __builtin_amdgcn_ballot_w64(newL <= newU && cLo == cHi) != 0ull;
                            if (maybe) {
                            const unsigned cw = (unsigned)(newU - newL);
#pragma unroll
                            for (int r = 0; r < RPL; ++r) {
                                const int i = b0[r] + lane;
                                const bool inr = (unsigned)(i - newL) <= cw;      // (no lane when the band is empty: both differences negative)
                                const unsigned long long rm = __builtin_amdgcn_ballot_w64(inr);
                                if (rm) {
                                    const int fl = (int)__builtin_ctzll(rm);
                                    const int v = __builtin_amdgcn_readlane(CS1[r], fl);
                                    const bool badS = __builtin_amdgcn_ballot_w64(inr && CS1[r] != v) != 0ull;
                                    const bool badID = __builtin_amdgcn_ballot_w64(inr && (CI1[r] != v || CD1[r] != v)) != 0ull;
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
                            if (vmin == vmax && !(fl & 1)) { conv_S = vmin; all3 = !(fl & 2); }   // (an empty band posts nothing: vmin > vmax)
                            }
                        }
                        if (all3 && prev_conv_s == conv_S && conv_S != -1) { converged = true; conv_value = prev_conv_s; convf = msp; }
                        prev_conv_s = conv_S;
                    }
                }
                {                                                                                  // :597-604
                    vlo2p = vlo1 + 1; vw2 = vw1;                // (an empty band keeps an unreachable low)
                    vlo1 = vL; vw1 = vwidth1; vwid1 = (int)vwidth1 + 1;
                    vL = max(max(newL, k + 2 - refLen), 0);
                    vU = min(newU + 1, qLen - 1);
                    Lk = __builtin_amdgcn_readfirstlane(vL);
                    Wk = __builtin_amdgcn_readfirstlane(vU - vL);
                    Lm63 = Lk - 63; W64 = Wk + 64;
                    vcur ^= parx; vprev ^= parx;
                }
#ifdef TWL_KERNEL_STAMPS
                {
                    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
                    st_slots += t_slots - t_head; st_bar += t_bar - t_slots; st_post += t_end - t_bar; st_n += 1;
                    st_act += (b0[0] <= Lk + Wk + 1 && b0[0] + 63 >= Lk) ? 1 : 0;
                    // timeline (tools/step_timeline.py): raw stamps of 96 diagonals of tile 4 from k = 600 (phase A) and from k = marker + 100 (phase C)
                    const int tlw = (k >= 600 && k < 696) ? k - 600 : ((k >= a.marker + 100 && k < a.marker + 196) ? 96 + k - a.marker - 100 : -1);
                    if (a.dbg && pair == 0 && tile == 4 && tlw >= 0 && lane == 0) {
                        long long *g = reinterpret_cast<long long *>(a.dbg + 16 * (size_t)a.n_pairs_total) + 256 + ((size_t)w * 192 + tlw) * 4;
                        g[0] = (long long)t_head; g[1] = (long long)t_slots; g[2] = (long long)t_bar; g[3] = (long long)t_end;
                    }
                }
#endif
                bool ended = false;
                if constexpr (CONV) {
                    if (converged) {                                                               // :607-612: max(0, max') > score at convergence
                        if (__builtin_amdgcn_ballot_w64(__int_as_float(max(__float_as_int(msp), 0)) > convf) != 0ull) { conv_logic = true; stop_tile(); ended = true; }
                    }
                }
                if (!ended) {
                    ++k;
                    // the hook of the group of 8 diagonals that has just ended (and of the marker diagonal: the words up to it are what the traceback reads)
                    if (k == kHook || (PH == 1 && k == marker + 1)) { hook(PHtag); kHook = (k & ~7) + 8; }
                    // stop conditions of the next diagonal (the reference tests them at its top, :323-338; not after the last one)
                    if (__builtin_expect((unsigned)Wk >= (unsigned)fcap, 0)) {
                        const int Uk = Lk + Wk;
                        if (k >= kEnd) {}                                                            // (that was the last diagonal)
                        else if (Lk > Uk) { tile_err = 1; stop_tile(); }                                 // band emptied by X-drop
                        else if (Uk - Lk + 1 > fLen) { tile_err = 2; stop_tile(); }                 // wider than fLen
                        else if ((Uk >> 6) - (Lk >> 6) >= NV) { tile_err = kErrOverflow; stop_tile(); }  // it really outgrew this window
                        else {
                            // a band this wide can need the successor of a block on the diagonal the block leaves it: advance before the activity test
#pragma unroll
                            for (int r = 0; r < RPL; ++r)
                                if (b0[r] + 63 < Lk) { while (b0[r] + 63 < Lk) b0[r] += 64 * NV; ring_addr(r, k); load_q(r); tbMust |= 1u << r; }
                        }
                    }
                }
            };

            if (steps_left < 0) { tile_err = 3; go = false; }
#ifdef TWL_KERNEL_STAMPS
            st_setup += __builtin_amdgcn_s_memtime() - st_tile0;
#endif
            {
                using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
                // The diagonals k .. kLim - 1 of one phase (round 6).  Special diagonals first, on the general step; then the plain step, whose whole loop
                // test is `k < kStop` (what ends the tile pulls kStop down).  The two marker diagonals (phase B) keep the general step.
                auto run = [&](auto PHtag, int kLim) __attribute__((always_inline)) {
                    constexpr int PH = decltype(PHtag)::value;
                    if constexpr (PH == 1) { while (go && k < kLim) step(PHtag, std::true_type{}); }
                    else {
                        while (go && spec && k < kLim) step(PHtag, std::true_type{});
                        kStop = go ? kLim : (int)0x80000000;
                        while (k < kStop) step(PHtag, std::false_type{});
                    }
                };
                const int kA = min(kEnd, marker - 1);
                // SPEC, while the tile runs on a guess: every 32 diagonals one lane looks whether the partner has published the true start;
                // the verdict goes through LDS so that every wave leaves the loop at the same diagonal
                auto poll = [&]() __attribute__((always_inline)) {
                    if (threadIdx.x == 0) {
                        const unsigned long long pv = team_load(&team[kTeamTruth + (tile & 3)]);
                        unsigned long long d = 0ull;
                        if ((unsigned)(pv >> 48) == (unsigned)(tile + 1)) {
                            const unsigned fl = (unsigned)(pv >> 32) & 0xFFFFu;
                            d = ((fl & (kTeamLast | kTeamErr)) || (int)((pv >> 16) & 0xFFFFu) != ref_idx || (int)(pv & 0xFFFFu) != qry_idx) ? 2ull : 1ull;
                        }
                        s_team[1] = d;
                        if (d == 1ull) atomicAdd(&team[kTeamStat + 1], 1ull);
                    }
                    __syncthreads();
                    const int d = __builtin_amdgcn_readfirstlane((int)s_team[1]);
                    __syncthreads();
                    // (no if / else if here: with that form the compiler stops treating the loop around it as uniform -- the loop counter and
                    // the band limits moved to vector registers and every branch of the step became an EXEC-mask branch)
                    confirmed = confirmed | (d == 1);
                    go = go & (d != 2);
                    aborted = aborted | (d == 2);
                };
                if constexpr (SPEC) {
                    while (go && k < kA) {
                        const int kc = confirmed ? kA : min(kA, (k | 31) + 1);
                        run(T0{}, kc);
                        if (go && !confirmed) poll();
                    }
                } else run(T0{}, kA);
                const int kB = min(kEnd, marker + 1);
                // (set here rather than with the tile: through phase A these are constants, not live registers; :306-308)
#pragma unroll
                for (int r = 0; r < RPL; ++r) { CS1[r] = -1; CI1[r] = kIB; CD1[r] = kDB; LCS2[r] = -1; }
                int sres = -1;
                if constexpr (SPEC) {          // the marker is not passed on a guess: the guess for the NEXT tile must rest on a true start
                    if (go && !confirmed && k < kB) {
                        sres = settle();
                        if (sres == 0) confirmed = true; else { go = false; aborted = true; }
                    }
                }
                run(T1{}, kB);
                if constexpr (SPEC) {
                    if (go && k == marker + 1 && k < kEnd) {     // both marker diagonals are done: tell the partner where the next tile probably starts
                        if (threadIdx.x == 0) {
                            const unsigned long long b0 = s_team[2], b1 = s_team[3];      // best cell of diagonal marker-1, marker
                            // a run of matches touches the diagonals of one parity only: the cell all paths will agree on lies on the
                            // marker diagonal or on the one before it (state 3, :520-524), whichever holds the better score
                            const bool onMarker = (unsigned)(b1 >> 32) >= (unsigned)(b0 >> 32);
                            const int gq = (int)((onMarker ? b1 : b0) & 0xFFFFFFFFu), gr = (onMarker ? marker : marker - 1) - gq;
#ifdef TWL_SPEC_DEBUG
                            printf("tile %d start %d %d: best(marker-1) key %x row %d, best(marker) key %x row %d -> guess +%d +%d\n", tile, ref_idx, qry_idx, (unsigned)(b0 >> 32), (int)(b0 & 0xFFFFFFFFu), (unsigned)(b1 >> 32), (int)(b1 & 0xFFFFFFFFu), gr, gq);
#endif
                            if ((b0 | b1) != 0ull && gr >= 0 && ref_idx + gr < 65536 && qry_idx + gq < 65536) {
                                team_store(&team[kTeamGuess + ((tile + 1) & 3)], team_word((unsigned)(tile + 2), 0u, (unsigned)(ref_idx + gr), (unsigned)(qry_idx + gq)));
                                atomicAdd(&team[kTeamStat + 0], 1ull);
                            }
                        }
                    }
                }
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
                run(T2{}, kCap);
                if (go && k < kEnd) { tile_err = kErrOverflow; go = false; }      // (never seen: tiles converge within ~1.5 markers)
                if constexpr (SPEC) {
                    if (aborted || !confirmed) {
                        if (sres < 0 || sres == 0) sres = settle();
                        if (sres == 1) { redo = true; confirmed = true; }          // run this tile again, from the true start
                        else if (sres == 2) { teamExit = true; last_tile = true; }   // the alignment is over
                        else confirmed = true;
                    } else sres = 0;
                    aborted = (sres != 0);
                }
            }
#ifdef TWL_KERNEL_STAMPS
            const unsigned long long st_exit0 = __builtin_amdgcn_s_memtime();
#endif
            if (!(SPEC && aborted)) {
            const int last_k = conv_logic ? k : k - 1;
            steps_left -= (long long)(last_k + 1);
            const unsigned tile_cells = (unsigned)__builtin_amdgcn_readfirstlane((int)vcells);
            const int lo1 = __builtin_amdgcn_readfirstlane(vlo1);

            cells += tile_cells;
            dbg_lastk = last_k; dbg_conv = conv_value; dbg_L = Lk; dbg_U = Lk + Wk;
            if (tile_err != 0) { err = tile_err; break; }
            // a profile entry outside fast_div's range: the IEEE-division kernel re-runs the pair (every wave saw different columns:
            // the verdict goes through LDS so that all of them leave together)
            if (PRESIM || SP == 1 || !denomOne) {
                if (guardBad) s_misc[4] = 1;
                __syncthreads();
                guardBad = __builtin_amdgcn_readfirstlane(s_misc[4]) != 0;
                if (guardBad) { err = kErrGuard; break; }
            }

            // a tile that ends before the marker leaves its last (partial) group of 8 diagonals unflushed
            if (tbPending) {
#pragma unroll
                for (int r = 0; r < RPL; ++r)
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(tb) + tbOff + (unsigned)(256 * (r * W + w))) = tbacc[r];
            }

            // ---- tile exit, :615-682 ----
            // after the loop (lo1, w1) is the band of diagonal last_k (rotated once more at its end)
            int conv_q = 0, conv_r = 0, tb_state = 0, start_k = 0;
            bool bad = false;
            if (!conv_logic && last_k >= marker) {                            // :633-635 needs CS[last_k][0]
                const int Llast = lo1;
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int b = b0[r];
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
#ifdef TWL_SPEC_DEBUG
            if (threadIdx.x == 0) printf("tile %d start %d %d: true +%d +%d state %d last_k %d conv_logic %d\n", tile, ref_idx, qry_idx, conv_r, conv_q, tb_state, last_k, (int)conv_logic);
#endif
            ref_idx += conv_r; qry_idx += conv_q;                             // :654-655
            if (R - ref_idx < 0 || Q - qry_idx < 0) { err = 3; break; }       // :659-668
            int tailDir = 0, tailLen = 0;
            if (ref_idx == R - 1 && qry_idx < Q - 1) { tailDir = 1; tailLen = Q - qry_idx - 1; last_tile = true; }   // :671-674
            if (qry_idx == Q - 1 && ref_idx < R - 1) { tailDir = 2; tailLen = R - ref_idx - 1; last_tile = true; }   // :675-678
            if (ref_idx == R - 1 && qry_idx == Q - 1) last_tile = true;       // :679
            if constexpr (SPEC) {       // the true start of the next tile (or: there is none), before the traceback: the partner is waiting for it
                if (threadIdx.x == 0)
                    team_store(&team[kTeamTruth + ((tile + 1) & 3)], team_word((unsigned)(tile + 2), last_tile ? kTeamLast : 0u, (unsigned)ref_idx, (unsigned)qry_idx));
            }

            __syncthreads();   // all traceback-pointer stores of this tile are complete and visible
            if (w == 0) {
                if constexpr (SPEC) {   // where this tile's segment goes: after the previous tile's, which the partner wrote
                    if (tile > 0) {
                        unsigned long long pv = 0ull;
                        if (lane == 0) pv = team_wait(&team[kTeamPos + (tile & 3)], (unsigned)(tile + 1));
                        pos = __builtin_amdgcn_readfirstlane((int)(pv & 0xFFFFFFFFu));
                        if (__builtin_amdgcn_readfirstlane((int)(pv >> 32)) == 0) { err = kErrOverflow; timedOut = true; }
                    }
                }
                int n = 0;
                {   // Traceback, :134-231, addressed by (diagonal, row) instead of a ragged offset.  One lane walks the pointers; a walk
                    // straight out of HBM costs a dependent L2 round trip per step, so the wave fetches the pointer words in patches of
                    // 64 rows x 16 groups of 8 diagonals (a path of matches crosses exactly that) into the reference ring, which is dead
                    // until the next tile, and the walk reads LDS.
                    constexpr int PG = 16;
                    static_assert(sizeof(s_ring) >= PG * 64 * sizeof(uint32_t), "traceback patch lives in the ring");
                    uint32_t *s_patch = reinterpret_cast<uint32_t *>(s_ring);
                    int kk2 = start_k, ii = conv_q, qi = conv_q, ri = conv_r, st = tb_state % 3;
                    const bool first = (tile == 0);
                    bool done = (kk2 < 0), walkBad = false;
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
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (one wave: its LDS operations complete in order)
                        if (lane == 0) {
                            for (;;) {
                                const int t = g0 - (kk2 >> 3), l = 63 - (i0 - ii);
                                if (t >= PG || l < 0) break;                      // the path left the patch
                                const uint32_t word = s_patch[t * 64 + l];
                                const int v = (int)((word >> (4 * (kk2 & 7))) & 0xFu);
                                // state bits 3 are never written by the DP (:548-557 has 0, 1, 2): a word that was skipped although the path runs through it -- the
                                // conditional store of the hook rests on the hook cadence and the band update -- reads as poison (TWL_KNOB_POISON_TB: 0xF) and ends
                                // the pair with the "bug" code instead of a wrong path
                                if ((v & 3) == 3) { walkBad = true; done = true; break; }
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
                                if constexpr (MT == 4) {      // the pair's scout keeps no path, only where it crosses every anti-diagonal (tile starts are read off it)
                                    int32_t *sp = a.mt_spath + (size_t)mtx * (size_t)a.mt_sp_pitch;
                                    const int dHere = (dir == 0) ? kk2 + 2 : kk2 + 1;      // the diagonal of the cell just left
                                    sp[dHere] = qi + ((dir != 2) ? 1 : 0);
                                    if (dir == 0 && dHere >= 1) sp[dHere - 1] = -1;
                                } else s_rev[n++] = (int8_t)dir;
                                if (kk2 < 0) { done = true; break; }
                                if (first && (ri < 0 || qi < 0)) { done = true; break; }
                                if (ii < 0) { done = true; break; }   // defensive: a pointer chain left the tile (never on valid data)
                            }
                        }
                        kk2 = __builtin_amdgcn_readfirstlane(kk2);
                        ii = __builtin_amdgcn_readfirstlane(ii);
                        done = __builtin_amdgcn_readfirstlane((int)done) != 0;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the walk's reads, before the next patch overwrites them
                    }
                    if (__builtin_amdgcn_readfirstlane((int)walkBad) != 0) err = 3;
                    if (lane == 0 && first && MT != 4) {
                        while (ri > -1) { s_rev[n++] = 2; ri--; }
                        while (qi > -1) { s_rev[n++] = 1; qi--; }
                    }
                }
                n = __builtin_amdgcn_readfirstlane(n);
                const int skip = (tile > 0) ? 1 : 0;                          // :98-102
                const int cnt = n - skip;
                if constexpr (MT == 1) {          // the segment and the record of this tile (read by the stitch launch)
                    if (err == 0 && cnt <= a.mt_segcap) {
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
                        for (int t = n - 2; t >= 0; --t) {      // (s_rev[n - 1] is the column of the start cell itself)
                            const int dir = s_rev[t];
                            r += (dir != 1) ? 1 : 0; q += (dir != 2) ? 1 : 0;
                            const int d = r + q;
                            if (d >= spLo && d <= dMax) sp[d] = q;
                        }
                    }
                } else if constexpr (MT == 4) {
                } else
                if (pos + cnt + tailLen > 2 * a.seq_len) { err = 3; }
                else {
                    for (int t = lane; t < cnt; t += 64) out[pos + t] = s_rev[n - 1 - skip - t];
                    for (int t = lane; t < tailLen; t += 64) out[pos + cnt + t] = (int8_t)tailDir;
                    pos += cnt + tailLen;
                }
                if (lane == 0) { s_misc[2] = err; s_misc[3] = pos; }
                if constexpr (SPEC) {
                    if (lane == 0 && err == 0 && !last_tile) team_store(&team[kTeamPos + ((tile + 1) & 3)], ((unsigned long long)(unsigned)(tile + 2) << 48) | (unsigned)pos);
                }
            }
            __syncthreads();
            err = __builtin_amdgcn_readfirstlane(s_misc[2] == 3 ? 3 : err);
            pos = __builtin_amdgcn_readfirstlane(s_misc[3]);
            iEnded = last_tile;
#ifdef TWL_KERNEL_STAMPS
            st_exit += __builtin_amdgcn_s_memtime() - st_exit0;
#endif
            if (err != 0) break;
            tile += SPEC ? 2 : 1;
            if constexpr (MT == 1 || MT == 2 || MT == 4) last_tile = true;      // one tile per job
            }      // (tile not thrown away)
            }      // (tile started)
        }
        if constexpr (SPEC) {
            if (err != 0 && (!teamExit || timedOut)) {       // this workgroup fails the pair: tell the partner (it may be waiting for a start)
                iEnded = true;
                if (threadIdx.x == 0) {
                    team_store(&team[kTeamTruth + ((tile + 1) & 3)], team_word((unsigned)(tile + 2), kTeamErr, 0u, 0u));
                    team_store(&team[kTeamTruth + ((tile + 2) & 3)], team_word((unsigned)(tile + 3), kTeamErr, 0u, 0u));      // (whichever tile it is about to ask for)
                }
            }
            if (R <= 0 || Q <= 0) iEnded = (role == 0);
        } else iEnded = true;

#ifdef TWL_KERNEL_STAMPS
        if (a.dbg && lane == 0 && pair == 0) {      // per wave of the workgroup that aligned pair 0: cycle sums per segment
            long long *g = reinterpret_cast<long long *>(a.dbg + 16 * (size_t)a.n_pairs_total) + 8 * w;
            g[0] = (long long)st_slots; g[1] = (long long)st_bar; g[2] = (long long)st_post; g[3] = (long long)st_n; g[4] = (long long)st_act;
            g[5] = (long long)(__builtin_amdgcn_s_memtime() - st_t0);
            g[6] = (long long)st_exit; g[7] = (long long)st_setup;
        }
#endif
        __syncthreads();
        if constexpr (MT == 2 || MT == 4) {
            if (threadIdx.x == 0 && err != 0 && a.mt_stat) atomicAdd(&a.mt_stat[2], 1ull);
        }
        if constexpr (MT == 1) {          // a tile job that ended with an error code leaves that as its record (read by the stitch launch)
            if (threadIdx.x == 0 && err != 0 && jobRef >= 0) {
                int32_t *rc = a.mt_rec + ((size_t)mtx * a.mt_slots + slot) * kMtRec;
                rc[1] = jobRef; rc[2] = jobQry; rc[9] = (int32_t)(unsigned)cells; rc[10] = err; rc[11] = WINDOW; rc[0] = 2;
            }
        }
        if constexpr (MT == 3) {
            if (threadIdx.x == 0 && !mtSkip) {
                int32_t *fr = a.mt_front + (size_t)mtx * 8;
                const unsigned long long c = cells + mtCells0;
                fr[1] = tile; fr[2] = ref_idx; fr[3] = qry_idx; fr[4] = pos; fr[6] = (int32_t)(unsigned)(c & 0xFFFFFFFFull); fr[7] = (int32_t)(unsigned)(c >> 32);
                fr[0] = suspended ? 1 : 2;
                if (a.mt_stat) { atomicAdd(&a.mt_stat[0], (unsigned long long)mtHits); atomicAdd(&a.mt_stat[1], (unsigned long long)mtInline); }
            }
            cells += mtCells0;
        }
        if (threadIdx.x == 0 && MT != 1 && MT != 2 && MT != 4 && !(MT == 3 && (suspended || mtSkip))) {
            if (iEnded || err != 0) {
                a.err[pair] = (int16_t)err;
                a.aln_len[pair] = (err == 0) ? pos : 0;
            }
            if constexpr (SPEC) atomicAdd(&a.cells[pair], cells);       // (zeroed by the host) both workgroups add their tiles
            else a.cells[pair] = cells;
            if (a.dbg) {
                int32_t *g = a.dbg + 16 * (size_t)pair;
                g[0] = tile; g[1] = dbg_lastk; g[2] = dbg_conv; g[3] = dbg_L; g[4] = dbg_U; g[5] = ref_idx; g[6] = qry_idx;
                g[7] = pos; g[8] = err; g[9] = (int)min(steps_left, 0x7fffffffll); g[10] = R; g[11] = Q;
            }
        }
        __syncthreads();   // keeps the single-lane block above out of the loop latch
    }
}

}  // namespace twl

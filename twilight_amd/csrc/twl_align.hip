// twilight_amd/csrc/twl_align.hip -- host side of libtwl_align (C ABI in include/twl_align.h).
//
// Owns device buffers, streams and the launch policy.  There is no CPU fallback: without a HIP
// device every entry point fails with TWL_ERR_HIP / TWL_ERR_NOT_INITIALIZED.
#include "../../include/twl_align.h"
#include "../../include/twl_level.h"
#include "level_kernels.hip.h"
#include "restore_kernels.hip.h"
#include "talco_kernel.hip.h"
#include "talco_nuc.hip.h"
#include "talco_global.hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace twl {
// Several byte fills in one launch (a level's DP call zeroes its outputs, work counters and tile tables: seven hipMemsetAsync before).
struct FillJob { void *p; unsigned long long bytes; unsigned int val; };
struct FillArgs { FillJob j[8]; int n; };
__global__ void __launch_bounds__(256) fill_kernel(FillArgs a)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    for (int r = 0; r < a.n; ++r) {
        const size_t nw = a.j[r].bytes >> 4;
        uint4 *w = reinterpret_cast<uint4 *>(a.j[r].p);
        const unsigned v = a.j[r].val;
        for (size_t i = t; i < nw; i += nt) w[i] = make_uint4(v, v, v, v);
        const size_t tail = a.j[r].bytes & 15;
        if (t < tail) reinterpret_cast<unsigned char *>(a.j[r].p)[(nw << 4) + t] = (unsigned char)v;
    }
}
// The per-pair results of a DP call (and the tile-parallel counters) into one host-visible block: [4 counters][n cells][n lengths][n codes]
__global__ void __launch_bounds__(256) collect_kernel(int n, const int16_t *err, const int32_t *aln_len, const unsigned long long *cells,
                                                      const unsigned long long *mt_stat, unsigned long long *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long *oc = out + 4;
    int32_t *ol = reinterpret_cast<int32_t *>(oc + n);
    int16_t *oe = reinterpret_cast<int16_t *>(ol + n);
    if (i < 4) out[i] = mt_stat ? mt_stat[i] : 0ull;
    if (i < n) { oc[i] = cells[i]; ol[i] = aln_len[i]; oe[i] = err[i]; }
    __threadfence_system();
}
}  // namespace twl

namespace {

thread_local std::string g_err;
std::mutex g_mu;

static bool dbg_on() { static const bool v = getenv("TWL_DEBUG") != nullptr; return v; }
#define TRACE(...) do { if (dbg_on()) { fprintf(stderr, "[twl trace] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                         \
            return TWL_ERR_HIP;                                                                                \
        }                                                                                                      \
    } while (0)

// Fast path: 8 waves x 2 row blocks per lane     -> 1024-row window (bands up to 961 wide), ref ring in LDS, 2 workgroups/CU.
// Wide path: 8 waves x 9 blocks                   -> 4608-row window (covers flen = 4096), ref columns from L2/HBM.
// Protein (P = 22): 8 waves x 1 block (512-row window, 96-byte columns in the LDS ring); wide path re-reads both columns per cell.

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return TWL_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return TWL_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Device {
    int id = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;                                               // row rewrites of small levels (twl_level_commit), beside the next level's work on `stream`
    hipEvent_t ev2[2] = {};                                                      // ... timed
    hipEvent_t ev[8] = {};                                                       // [6], [7]: the write-back of a level (twl_level_commit, read later by twl_level_timing)
    int num_cu = 0;
    Buf cols, tb, cells, queue, items, errs, dbg;
    Buf gtb;                                                                     // scratch of the global-memory kernel (DP rows + pointer bytes of its pairs): its own buffer, handed back after the stage when large
    Buf sim, sim_off, blk_off, m24;                                              // precomputed protein scores (matrix mode 4)
    Buf team;                                                                    // mailboxes of the speculative tile start
    Buf mt_chain, mt_rec, mt_seg, mt_spath, mt_stat, mt_jobs, mt_anchor;                    // tile-parallel alignment (talco_nuc.hip.h, MT kernels)
    Buf simdump;                                                                 // twl_dp_column_scores: [Q][R] scores written by the DUMP kernels
    bool dump_on = false;
    std::vector<int32_t> dbg_host;
    std::vector<int32_t> mt_jobs_host;
    Buf gc_zero;                          // [pair] flags of a call with pairs of both gap-character kinds (run_device)
    std::vector<uint8_t> gc_zero_host;
    std::vector<int16_t> last_err;                                               // error codes of the last run_device call, as read back by it
    void *probe_h = nullptr; size_t probe_cap = 0;                               // pinned: error codes of a level's sample (run_device, Throughput)
    bool corridor_lost = false; int corridor_last_n = 0;                         // protein: a pair of an earlier level of this pass left the corridor of the precomputed scores (the later levels score the whole matrix)
    int small_state = 0, small_last_n = 0;                                       // plan_nucleotide: what the levels of short pairs of this pass found of the 512-row throughput window (1 fits, -1 outgrown), and the pairs of the last such level
    int live_stores = 0;                                                         // twl_store handles alive on this device (twl_level.h); guarded by mu
    void *comm = nullptr;                                                        // ncclComm_t of a sharded run (twl_comm_init)
    int comm_world = 0, comm_rank = 0;
    Buf comm_send, comm_recv;                                                    // staging of twl_comm_all_gather_host
    int wide_streak = 0;                                                         // consecutive small calls whose pairs all outgrew the fast window (run_device: wideFirst)
    int last_wide_pct = 0, wide_calls = 0;                                       // share of the last narrow-first call's pairs that went on to the wide window; calls started wide since
    int mt_launch = 0;                                                           // launches of the tile-parallel level in flight (work counter index)
    char kname[160] = {0};                                                       // the kernel of the first DP launch of the call in flight
    Buf h2d_freq, h2d_gop, h2d_gex, h2d_len, h2d_num, d_aln, d_alnlen, d_err;   // staging for the host form
    twl_stats stats{};
    std::vector<uint64_t> pair_cells;
    twl::FillArgs fills{};                                                        // byte fills queued for ONE launch in front of the next kernel (queue_fill / flush_fills)
    char *res_h = nullptr;                                                       // pinned host block the results of a call come back in (collect_kernel writes it)
    size_t res_cap = 0;
    std::vector<int32_t> last_alnlen;                                            // path lengths of the last run_device call, as read back by it
    std::mutex mu;
};

std::vector<Device *> g_devs;
bool g_init = false;

int find_dev(int device, Device **out)
{
    for (auto *d : g_devs) if (d->id == device) { *out = d; return TWL_OK; }
    g_err = "device not selected in twl_init";
    return TWL_ERR_BAD_ARGUMENT;
}

// Byte fills queued in front of the next kernel of the device's stream: one launch for all of them (16-byte aligned pointers).
int flush_fills(Device *d, hipStream_t st)
{
    if (d->fills.n == 0) return TWL_OK;
    unsigned long long most = 0;
    for (int r = 0; r < d->fills.n; ++r) most = std::max(most, d->fills.j[r].bytes);
    const unsigned blocks = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>((most / 16 + 255) / 256, (unsigned long long)d->num_cu * 8));
    hipLaunchKernelGGL(twl::fill_kernel, dim3(blocks), dim3(256), 0, st, d->fills);
    d->fills.n = 0;
    HIP_TRY(hipGetLastError());
    return TWL_OK;
}
int queue_fill(Device *d, hipStream_t st, void *p, size_t bytes, unsigned char v)
{
    if (!bytes) return TWL_OK;
    if (((size_t)p & 15) != 0) { HIP_TRY(hipMemsetAsync(p, v, bytes, st)); return TWL_OK; }
    if (d->fills.n == 8) { const int rc = flush_fills(d, st); if (rc) return rc; }
    d->fills.j[d->fills.n++] = twl::FillJob{p, (unsigned long long)bytes, 0x01010101u * v};
    return TWL_OK;
}
#define FILL_TRY(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)

int check_params(const twl_params *p)
{
    if (!p) { g_err = "params is null"; return TWL_ERR_BAD_ARGUMENT; }
    if (p->P != 6 && p->P != 22) { g_err = "profile width P must be 6 (nucleotide) or 22 (protein)"; return TWL_ERR_UNSUPPORTED; }
    if (p->marker < 2 || p->marker > TWL_MAX_MARKER) { g_err = "marker outside [2, TWL_MAX_MARKER]"; return TWL_ERR_UNSUPPORTED; }
    // flen is only a cap on the anti-diagonal width (TALCO-XDrop.cpp:258,331-338): the deferred pass raises it to min(R, Q)
    // (alignment-cpu.cpp:116-129).  Any value is accepted; what is limited is the band the widest kernel can hold (4608 rows).
    if (p->flen < 1) { g_err = "flen < 1"; return TWL_ERR_UNSUPPORTED; }
    if (p->xdrop < 0) { g_err = "xdrop < 0"; return TWL_ERR_BAD_ARGUMENT; }
    return TWL_OK;
}

#include "twl_knobs.inc.hip"
#include "twl_launch.inc.hip"

#include "twl_policy.inc.hip"


// Device-resident core.  len/num are needed on the host for cost ordering (they are tiny).
int run_device(Device *d, hipStream_t st, const twl_params *p, int32_t n_pairs, int32_t seq_len, const float *d_freq,
               const float *d_gop, const float *d_gex, const int32_t *d_len, const int32_t *d_num, int8_t *d_aln,
               int32_t *d_alnlen, int16_t *d_err, const int32_t *h_len, const float *d_packed = nullptr, bool qry_onehot = false,
               const uint8_t *h_gc_zero = nullptr, int shape = 0)
{
    // h_gc_zero: optional [n_pairs], 1 = the pair's gapCharScore is 0 whatever p->gap_char says (the reference decides it per pair,
    // alignment-cpu.cpp:88; one launch then takes the pairs of both kinds -- the top levels of a 100 000-leaf tree hold a few of each)
    // shape: what the caller knows about EVERY pair that runs: 2 = single sequences on both sides (no gap letters, denominators of 1): the throughput kernels
    // then run a step without the per-block tests (talco_lean_kernel, SP 1)
    // qry_onehot: every query row of every pair of this call has at most one non-zero letter (single sequences: the device-resident
    // level path knows, it built the profiles) -- the nucleotide kernels then take the four-product form of the column score
    // d_packed: the level's columns already in the packed [P+2] layout (device-resident level path); no packing pass then
    HIP_TRY(hipSetDevice(d->id));
    d->stats = twl_stats{};
    d->kname[0] = 0;
    d->fills.n = 0;
    d->last_err.clear();
    d->last_alnlen.clear();
    d->pair_cells.assign((size_t)n_pairs, 0);
    if (n_pairs == 0) return TWL_OK;

    int rc;
    const size_t n_cols = (size_t)n_pairs * 2 * (size_t)seq_len;
    const bool prot = (p->P == 22);
    if (!d_packed && (rc = d->cols.ensure(n_cols * (size_t)(p->P + 2) * sizeof(float)))) return rc;
    if ((rc = d->cells.ensure((size_t)n_pairs * sizeof(unsigned long long)))) return rc;
    if ((rc = d->queue.ensure(64))) return rc;
    if ((rc = d->items.ensure((size_t)n_pairs * sizeof(int32_t)))) return rc;

    // cost order: longest first (LPT) so the persistent workgroups finish together
    std::vector<int32_t> len_host;
    if (!h_len) {
        len_host.resize((size_t)n_pairs * 2);
        HIP_TRY(hipMemcpyAsync(len_host.data(), d_len, len_host.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h_len = len_host.data();
    }
    std::vector<int32_t> order((size_t)n_pairs);
    std::iota(order.begin(), order.end(), 0);
    // (pairs with an empty side -- masked out by the caller, or really empty -- come last and are not launched at all: what the geometry
    // is chosen by is the number of pairs that run, e.g. a rank's share of a level)
    auto live = [&](int32_t x) { return h_len[2 * x] > 0 && h_len[2 * x + 1] > 0; };
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        if (live(x) != live(y)) return live(x);
        return (int64_t)h_len[2 * x] + h_len[2 * x + 1] > (int64_t)h_len[2 * y] + h_len[2 * y + 1];
    });
    int32_t n_run = 0;
    for (int32_t n = 0; n < n_pairs; ++n) n_run += live(n) ? 1 : 0;
    uint64_t nominal = 0;
    for (int32_t n = 0; n < n_pairs; ++n) nominal += (uint64_t)std::max(0, h_len[2 * n]) * (uint64_t)std::max(0, h_len[2 * n + 1]);
    HIP_TRY(hipMemcpyAsync(d->items.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));

    TRACE("run_device n_pairs=%d seq_len=%d", n_pairs, seq_len);
    HIP_TRY(hipEventRecord(d->ev[0], st));
    if (!d_packed) {
        const int threads = 256;
        const int blocks = (int)std::min<size_t>((n_cols + threads - 1) / threads, (size_t)d->num_cu * 8);
        if (prot) hipLaunchKernelGGL(twl::pack_kernel<22>, dim3(std::max(blocks, 1)), dim3(threads), 0, st, d_freq, d_gop, d_gex, (float *)d->cols.p, n_cols);
        else hipLaunchKernelGGL(twl::pack_kernel<6>, dim3(std::max(blocks, 1)), dim3(threads), 0, st, d_freq, d_gop, d_gex, (float *)d->cols.p, n_cols);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(d->ev[1], st));
    if (dbg_on()) { HIP_TRY(hipStreamSynchronize(st)); TRACE("pack done"); }

    twl::KArgs a{};
    a.cols = d_packed ? d_packed : (const float *)d->cols.p;
    a.len = d_len; a.num = d_num;
    a.aln = d_aln; a.aln_len = d_alnlen; a.err = d_err;
    a.cells = (unsigned long long *)d->cells.p;
    a.queue = (int32_t *)d->queue.p;
    a.seq_len = seq_len;
    a.n_pairs_total = n_pairs;
    a.gap_open = p->gap_open; a.gap_extend = p->gap_extend; a.gap_char = p->gap_char;
    a.gc_zero = nullptr;
    if (h_gc_zero && p->gap_char != 0.0f && std::any_of(h_gc_zero, h_gc_zero + n_pairs, [](uint8_t z) { return z != 0; })) {
        if ((rc = d->gc_zero.ensure((size_t)n_pairs))) return rc;
        d->gc_zero_host.assign(h_gc_zero, h_gc_zero + n_pairs);       // (kept with the device: the upload is asynchronous)
        HIP_TRY(hipMemcpyAsync(d->gc_zero.p, d->gc_zero_host.data(), (size_t)n_pairs, hipMemcpyHostToDevice, st));
        a.gc_zero = (const uint8_t *)d->gc_zero.p;
    }
    a.xdrop = p->xdrop; a.flen = p->flen; a.marker = p->marker;
    a.step_slack = 1 << 16;
    const bool want_dbg = dbg_on();
    a.dbg = nullptr;
    if (want_dbg) {
        if ((rc = d->dbg.ensure((size_t)n_pairs * 16 * sizeof(int32_t) + 2048 + 16 * 192 * 32))) return rc;     // per-pair records, then the stamp build's sums and timeline
        HIP_TRY(hipMemsetAsync(d->dbg.p, 0xff, (size_t)n_pairs * 16 * sizeof(int32_t), st));
        a.dbg = (int32_t *)d->dbg.p;
    }
    { const int ms = p->P - 1; for (int l = 0; l < ms; ++l) for (int m = 0; m < ms; ++m) a.M[ms * l + m] = p->matrix[ms * l + m]; }

    // the pairs that do not run: path length 0, errorType 0, no cells
    FILL_TRY(queue_fill(d, st, d_alnlen, (size_t)n_pairs * sizeof(int32_t), 0));
    FILL_TRY(queue_fill(d, st, d_err, (size_t)n_pairs * sizeof(int16_t), 0));
    FILL_TRY(queue_fill(d, st, d->cells.p, (size_t)n_pairs * sizeof(unsigned long long), 0));

    int grid = 0, window = 0;
    bool protSmall = false;       // protein, first stage on the 512-row kernel
    int statMode = -1, statSpec = 0;
    bool ranMt = false, leanMid = false, startedWide = false, thr768 = false, thr512 = false;
    bool usedCorridor = false;            // protein: the scores of this call were precomputed in a corridor only
    bool smallTiles = false;              // the tile jobs of a tile-parallel launch of this call ran on the 512-row window
    bool probed = false;                  // the level's own sample kept the level off the 512-row window (and set the memory of it)
    int from512Pairs = -1;                // pairs of a 512-row throughput launch that outgrew it (-1: no such launch)
    const int32_t *items = (const int32_t *)d->items.p;
    auto launch_wide = [&](const int32_t *it, int n_it, int *g, int *w) {
        return prot ? launch_dp<22, 8, 9, false, false, false>(d, st, a, it, n_it, 1, g, w)
                    : launch_dp<6, 8, 9, false, false, true>(d, st, a, it, n_it, 1, g, w);
    };
    if (n_run == 0) rc = TWL_OK;      // nothing to align in this call
    else if (g_force_global) {
        rc = prot ? launch_global<22>(d, st, a, items, n_run, seq_len, &grid, &window) : launch_global<6>(d, st, a, items, n_run, seq_len, &grid, &window);
        snprintf(d->kname, sizeof d->kname, "talco_global_kernel<%d>", prot ? 22 : 6);
    }
    else if (prot) {
        // default: sparse score loop over the non-zero letters of the reference column (matrix mode 3, bit-identical to the dense loop)
        // TWL_KNOB_PROT_MODE: auto | dense | sparse | presim | r1 (round-1 kernels) | lean_sparse | lean_presim
        static const char *const kProtModes[] = {"auto", "dense", "sparse", "presim", "r1", "lean_sparse", "lean_presim"};
        const std::string pcs = kProtModes[std::max(0, std::min(6, g_prot_mode))];
        // fast_div's guard (talco_nuc.hip.h): non-zero scores within [2^-10, 2^10]
        bool divOk = true;
        auto inRange = [](float x) { const float ax = std::fabs(x); return x == 0.0f || (ax >= 0.0009765625f && ax <= 1024.0f); };
        for (int t = 0; t < 441; ++t) divOk = divOk && inRange(a.M[t]);
        divOk = divOk && inRange(p->gap_char);
        const bool lean = divOk && (pcs == "auto" || pcs == "lean_sparse" || pcs == "lean_presim");
        if (pcs == "r1") rc = launch_dp<22, 8, 1, false, true, true>(d, st, a, items, n_run, 0, &grid, &window);
        else if (pcs == "dense") rc = launch_dp<22, 8, 2, false, true, true>(d, st, a, items, n_run, 0, &grid, &window);
        else {
            // Few pairs (upper tree levels): the serial diagonal chain of each pair is what costs, and most of its instructions are the
            // column score.  Scores do not depend on the DP state, so the otherwise idle CUs compute them for the whole R x Q matrix
            // first (score_matrix_kernel, same arithmetic) and the DP kernel only loads them (matrix mode 4).
            size_t simFloats = 0;
            std::vector<long long> off((size_t)n_pairs, 0);
            std::vector<int32_t> blk((size_t)n_pairs + 1, 0);
            for (int32_t t = 0; t < n_run; ++t) {
                const int32_t pr = order[t];
                const long long R = std::max(0, h_len[2 * pr]), Q = std::max(0, h_len[2 * pr + 1]);
                off[pr] = (long long)simFloats;
                const long long pitch = (Q + 63) & ~63ll;
                const bool live = R > 0 && Q > 0;
                simFloats += live ? (size_t)((R + Q) * pitch) : 0;
                blk[t + 1] = blk[t] + (live ? (int32_t)(((R + Q - 1 + 63) / 64) * ((Q + 63) / 64)) : 0);
            }
            int32_t maxLenP = 0;
            for (int32_t t = 0; t < 2 * n_pairs; ++t) maxLenP = std::max(maxLenP, h_len[t]);
            const bool few = n_run <= std::max(1, d->num_cu / 2);      // measured break-even vs the sparse in-kernel path: ~150 pairs of 2 kaa
            const bool fits = simFloats * sizeof(float) <= ((size_t)16 << 30) && blk[n_run] > 0;
            if ((rc = d->m24.ensure(21 * 24 * sizeof(float)))) return rc;
            std::vector<float> m24(21 * 24, 0.0f);
            for (int l = 0; l < 21; ++l) for (int m = 0; m < 21; ++m) m24[24 * l + m] = a.M[21 * l + m];
            HIP_TRY(hipMemcpyAsync(d->m24.p, m24.data(), m24.size() * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));      // m24 goes out of scope
            // CUs/2 < pairs <= CUs: speculative teams of the 512-row geometry, two workgroups per CU, on precomputed scores (as the nucleotide
            // path does with its throughput geometry)
            const bool sharedSpec = lean && pcs == "auto" && !few && n_run <= d->num_cu && maxLenP <= 65535 && fits && !d->dump_on && !g_no_spec;
            const bool presim = (pcs == "presim" || pcs == "lean_presim" || (pcs == "auto" && few) || sharedSpec) && fits && !d->dump_on;
            statMode = presim ? 4 : 3;
            if (presim) {
                if ((rc = d->sim.ensure(simFloats * sizeof(float)))) return rc;
                if ((rc = d->sim_off.ensure(off.size() * sizeof(long long)))) return rc;
                if ((rc = d->blk_off.ensure(blk.size() * sizeof(int32_t)))) return rc;
                HIP_TRY(hipMemcpyAsync(d->sim_off.p, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpyAsync(d->blk_off.p, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));      // the host vectors above go out of scope
                twl::ScoreArgs sa{};
                sa.cols = a.cols; sa.len = d_len; sa.num = d_num; sa.items = items; sa.blk_off = (const int32_t *)d->blk_off.p;
                sa.n_items = n_run; sa.seq_len = seq_len; sa.gap_char = p->gap_char; sa.gc_zero = a.gc_zero; sa.M24 = (const float *)d->m24.p;
                sa.sim = (float *)d->sim.p; sa.sim_off = (const long long *)d->sim_off.p;
                // (the lean kernels check what they read; the round-1 kernel behind `dense` does not.  A pair that leaves the corridor is re-run by the kernel that scores
                //  in line, at a hundred times what the corridor saved on it: the first such pair takes the rest of the pass off the corridor -- bands widen and paths
                //  wander up the tree -- and a level LARGER than the one before it is another pass or family and starts afresh, as the 512-row window does)
                if (n_run > d->corridor_last_n) d->corridor_lost = false;
                d->corridor_last_n = n_run;
                sa.corridor = (lean && !d->corridor_lost) ? g_prot_corridor : 0;
                usedCorridor = sa.corridor > 0;
                FILL_TRY(flush_fills(d, st));
                hipLaunchKernelGGL(twl::score_matrix_kernel<22>, dim3((unsigned)blk[n_run]), dim3(256), 0, st, sa);
                HIP_TRY(hipGetLastError());
                a.sim = (const float *)d->sim.p;
                a.sim_off = (const long long *)d->sim_off.p;
                statSpec = sharedSpec ? 2 : ((lean && 2 * n_run <= d->num_cu && maxLenP <= 65535 && !g_no_spec) ? 1 : 0);
                long long sumLenP = 0;
                for (int32_t t = 0; t < n_run; ++t) sumLenP += (long long)h_len[2 * order[t]] + h_len[2 * order[t] + 1];
                // tile-parallel (talco_nuc.hip.h, MT kernels) on the precomputed scores: pairs of 2 kaa have 4-5 tiles each
                const bool mtOkP = lean && pcs == "auto" && n_run <= g_mt_max_pairs && n_run <= d->num_cu && p->marker >= g_mt_min_marker && sumLenP >= 3ll * p->marker * n_run;
                if (mtOkP) { rc = launch_mt<22, 4, 1>(d, st, a, items, order, n_run, h_len, &grid, &window); statSpec = 3; ranMt = true; protSmall = false; }
                else if (sharedSpec) { rc = launch_lean<22, 8, 1, 4, 4, true>(d, st, a, items, n_run, &grid, &window); protSmall = true; }
                else if (lean && 2 * n_run <= d->num_cu && maxLenP <= 65535 && !g_no_spec) rc = launch_lean<22, 16, 1, 4, 1, true>(d, st, a, items, n_run, &grid, &window);
                else if (lean) rc = launch_lean<22, 16, 1, 4, 1>(d, st, a, items, n_run, &grid, &window);
                else rc = launch_dp<22, 8, 2, false, true, true, 1, 4>(d, st, a, items, n_run, 0, &grid, &window);
            } else if (d->dump_on) {      // twl_dp_column_scores: the sparse in-kernel score loop, every visited cell written out
                if (!lean || n_run != 1) { g_err = "score dump: one pair, matrix within the fast-division range"; return TWL_ERR_UNSUPPORTED; }
                rc = launch_lean<22, 16, 1, 3, 1, false, true>(d, st, a, items, n_run, &grid, &window);
            } else if (lean && n_run > d->num_cu) {
                // more pairs than CUs: the 512-row window (8 waves, one block each; protein bands of 2 kaa pairs are ~270 rows wide, ~400
                // at most) keeps the ring at 61 KB, so two workgroups share a CU like in the nucleotide throughput kernel; a pair
                // whose band outgrows it goes to the 1024-row kernel below
                rc = launch_lean<22, 8, 1, 3, 4>(d, st, a, items, n_run, &grid, &window);
                protSmall = true;
            } else if (lean) {
                rc = launch_lean<22, 16, 1, 3, 1>(d, st, a, items, n_run, &grid, &window);
            } else {
                rc = launch_dp<22, 8, 2, false, true, true, 1, 3>(d, st, a, items, n_run, 0, &grid, &window);
            }
        }
    }
    else {
        NucFacts nf;
        nf.n_run = n_run; nf.num_cu = d->num_cu; nf.marker = p->marker; nf.M = a.M; nf.gap_char = p->gap_char; nf.qry_onehot = qry_onehot; nf.shape = shape; nf.dump = d->dump_on;
        nf.wide_streak = d->wide_streak; nf.last_wide_pct = d->last_wide_pct; nf.wide_calls = d->wide_calls; nf.small_state = (n_run > d->small_last_n) ? 0 : d->small_state; nf.h_len = h_len; nf.order = order.data();
        const NucPlan pl = plan_nucleotide(nf, current_knobs());
        if (pl.first == NucFirst::Throughput && (pl.small || pl.held_back)) { if (n_run > d->small_last_n) d->small_state = 0; d->small_last_n = n_run; }
        const int mm = pl.mm;
        const bool mm5 = pl.mm5;
        statMode = mm5 ? 5 : mm;
        leanMid = pl.lean && mm == 2;
        TRACE("plan: %s%s, matrix mode %d%s", nuc_first_name(pl.first), pl.small ? " (512-row window, five workgroups per CU)" : "", mm, mm5 ? " (one-letter query rows)" : "");
        switch (pl.first) {
        case NucFirst::Dump:      // twl_dp_column_scores: the same kernel code with the score of every visited cell written out
            if (!pl.lean || n_run != 1) { g_err = "score dump: one pair, matrix within the fast-division range"; return TWL_ERR_UNSUPPORTED; }
            if (mm5) rc = launch_lean<6, 16, 1, 5, 1, false, true>(d, st, a, items, n_run, &grid, &window);
            else if (mm == 2) rc = launch_lean<6, 16, 1, 2, 1, false, true>(d, st, a, items, n_run, &grid, &window);
            else if (mm == 1) rc = launch_lean<6, 16, 1, 1, 1, false, true>(d, st, a, items, n_run, &grid, &window);
            else rc = launch_lean<6, 16, 1, 0, 1, false, true>(d, st, a, items, n_run, &grid, &window);
            break;
        case NucFirst::WideMt:
            rc = launch_mt<6, 2, 3, true, 4>(d, st, a, items, order, n_run, h_len, &grid, &window);
            statSpec = 3; ranMt = true; startedWide = true;
            break;
        case NucFirst::Mt:        // few pairs of many tiles each: all tiles of all pairs side by side from predicted starts (talco_nuc.hip.h, MT kernels)
            smallTiles = nf.small_state > 0;      // (the throughput levels of this pass fitted the 512-row window: so do the tiles of their pairs' descendants, until they do not)
            rc = launch_mt<6, 2, 3, false, 4>(d, st, a, items, order, n_run, h_len, &grid, &window, smallTiles, pl.sp);
            statSpec = 3; ranMt = true;
            break;
        case NucFirst::SpecShared:
            rc = launch_lean<6, 8, 2, 2, 4, true>(d, st, a, items, n_run, &grid, &window);
            statMode = 2; statSpec = 2;
            break;
        case NucFirst::Spec16:
            rc = mm5 ? launch_lean<6, 16, 1, 5, 1, true>(d, st, a, items, n_run, &grid, &window) : launch_lean<6, 16, 1, 2, 1, true>(d, st, a, items, n_run, &grid, &window);
            statSpec = 1;
            break;
        case NucFirst::Few16:
            if (mm5) rc = launch_lean<6, 16, 1, 5, 1>(d, st, a, items, n_run, &grid, &window);
            else if (mm == 2) rc = launch_lean<6, 16, 1, 2, 1>(d, st, a, items, n_run, &grid, &window);
            else if (mm == 1) rc = launch_lean<6, 16, 1, 1, 1>(d, st, a, items, n_run, &grid, &window);
            else rc = launch_lean<6, 16, 1, 0, 1>(d, st, a, items, n_run, &grid, &window);
            break;
        case NucFirst::Throughput: {
            // Default matrix structure (modes 2 and 5): 4 waves x 3 blocks, a 768-row window, FOUR workgroups per CU (round 4: the same 16 waves per CU as
            // 8 waves x 2 blocks twice, but four independent anti-diagonal chains per SIMD instead of two, a barrier of four waves instead of eight, and no
            // first products kept per row -- 2048 pairs of 10 kbp 120 -> 95 ms, a leaf level 96 -> 66 ms, tools/exp_thr.py); a pair whose band outgrows 640
            // rows re-runs on 8 waves x 2 blocks (1024 rows) below.
            const int bulk = pl.bulk, tail = pl.tail;
#if defined(TWL_EXP_THR_W)      // geometry experiments (tools/exp_thr.py on cross-compiled variants): waves, blocks per wave, waves per SIMD of the throughput launch
            if (mm5) rc = launch_lean<6, TWL_EXP_THR_W, TWL_EXP_THR_RPL, 5, TWL_EXP_THR_MINW>(d, st, a, items, bulk, &grid, &window);
            else if (mm == 2) rc = launch_lean<6, TWL_EXP_THR_W, TWL_EXP_THR_RPL, 2, TWL_EXP_THR_MINW>(d, st, a, items, bulk, &grid, &window);
#else
            bool small = pl.small;
            int done = 0;                 // pairs of the bulk the sample launch took
            if (pl.probe && bulk >= 8 * d->num_cu) {
                // the sample: one pair per CU, every (bulk / CUs)-th of the cost order, moved to the front of the launch order
                const int S = d->num_cu;
                std::vector<int32_t> front, rest;
                front.reserve((size_t)S); rest.reserve((size_t)bulk);
                for (int t = 0, nextPick = 0, j = 0; t < bulk; ++t) {
                    if (j < S && t == nextPick) { front.push_back(order[t]); ++j; nextPick = (int)((long long)j * bulk / S); }
                    else rest.push_back(order[t]);
                }
                std::copy(front.begin(), front.end(), order.begin());
                std::copy(rest.begin(), rest.end(), order.begin() + (std::ptrdiff_t)front.size());
                HIP_TRY(hipMemcpyAsync(d->items.p, order.data(), (size_t)bulk * sizeof(int32_t), hipMemcpyHostToDevice, st));
                done = (int)front.size();
                rc = launch_thr<4, 2, 5>(d, st, a, items, done, &grid, &window, mm5, pl.sp);
                if (rc) return rc;
                if ((size_t)n_pairs * sizeof(int16_t) > d->probe_cap) {
                    if (d->probe_h) (void)hipHostFree(d->probe_h);
                    d->probe_h = nullptr; d->probe_cap = 0;
                    HIP_TRY(hipHostMalloc((void **)&d->probe_h, (size_t)n_pairs * sizeof(int16_t) * 2, hipHostMallocDefault));
                    d->probe_cap = (size_t)n_pairs * sizeof(int16_t) * 2;
                }
                HIP_TRY(hipMemcpyAsync(d->probe_h, d_err, (size_t)n_pairs * sizeof(int16_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                int outgrew = 0;
                for (int32_t pr : front) outgrew += ((const int16_t *)d->probe_h)[pr] == twl::kErrOverflow ? 1 : 0;
                small = outgrew * 100 <= 1 * done;
                d->small_state = small ? 1 : -1;          // (the levels that follow in this pass do as this one did)
                probed = !small;                          // (a sample that said yes leaves the verdict to the whole level, below)
                TRACE("sample of %d pairs on the 512-row window: %d outgrew it -> the level runs on %d rows", done, outgrew, small ? 512 : 768);
                d->kname[0] = 0;                          // (the level's kernel is the one the rest runs on)
            }
            if (small && (mm5 || mm == 2)) rc = launch_thr<4, 2, 5>(d, st, a, items + done, bulk - done, &grid, &window, mm5, pl.sp);
            else if (mm5 || mm == 2) rc = launch_thr<4, 3, 4>(d, st, a, items + done, bulk - done, &grid, &window, mm5, pl.sp);
#endif
            else if (mm == 1) rc = launch_lean<6, 8, 2, 1, 2>(d, st, a, items, bulk, &grid, &window);
            else rc = launch_lean<6, 8, 2, 0, 2>(d, st, a, items, bulk, &grid, &window);
#if defined(TWL_EXP_THR_W)
            const bool small = false;
#endif
            thr768 = pl.four && !small;      // (after a sample that said no, its own pairs that outgrew 512 rows go straight on to the 1024-row window with the rest's)
            thr512 = small;
            if (!rc && tail > 0) {
                const std::vector<int32_t> tailOrder(order.begin() + bulk, order.begin() + n_run);
                int g2 = 0, w2 = 0;
                // (one-letter query rows: the tiles too take the four-product form of the column score)
                smallTiles = small && !probed;
                rc = mm5 ? launch_mt<6, 5, 3, false, 4>(d, st, a, items + bulk, tailOrder, tail, h_len, &g2, &w2, smallTiles, pl.sp)
                         : launch_mt<6, 2, 3, false, 4>(d, st, a, items + bulk, tailOrder, tail, h_len, &g2, &w2, smallTiles, pl.sp);
                ranMt = true;
            }
            break;
        }
        default:                  // scores outside the fast division's range: the round-1 kernels (IEEE division)
            if (mm == 2) rc = launch_dp<6, 8, 2, false, true, true, 4, 2>(d, st, a, items, n_run, 0, &grid, &window);
            else if (mm == 1) rc = launch_dp<6, 8, 2, false, true, true, 4, 1>(d, st, a, items, n_run, 0, &grid, &window);
            else rc = launch_dp<6, 8, 2, false, true, true, 4, 0>(d, st, a, items, n_run, 0, &grid, &window);
        }
    }
    if (rc) return rc;
    FILL_TRY(flush_fills(d, st));      // (nothing ran: the outputs are still to be zeroed)
    HIP_TRY(hipEventRecord(d->ev[2], st));
    d->stats.n_launches = 1;
    d->stats.grid = grid;
    d->stats.window = window;
    d->stats.matrix_mode = statMode;
    d->stats.speculative = statSpec;
    memcpy(d->stats.kernel, d->kname, sizeof d->stats.kernel);

    // Pairs whose band outgrew a window are re-run, bit-identically, by the next stage: 1024-row fast window ->
    // 2048-row window (16 waves x 2 blocks, LDS ring) -> 4608-row window (covers flen = 4096; columns from L2/HBM).
    // (error codes, band cells and the tile-parallel counters come back in ONE synchronisation when nothing has to be re-run -- the common case)
    std::vector<int16_t> h_err((size_t)n_pairs);
    std::vector<unsigned long long> cells((size_t)n_pairs);
    unsigned long long mtStat[4] = {0, 0, 0, 0};
    bool redoMt = false;      // a re-run went through the tile-parallel path (launch_mt, WIDE)
    int widePairs = 0;        // pairs that went on to the wide window
    // (one small kernel writes them into a pinned host block: three device-to-host copies into pageable memory before)
    const size_t resBytes = 4 * sizeof(unsigned long long) + (size_t)n_pairs * (sizeof(unsigned long long) + sizeof(int32_t) + sizeof(int16_t));
    if (resBytes > d->res_cap) {
        if (d->res_h) (void)hipHostFree(d->res_h);
        d->res_h = nullptr; d->res_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&d->res_h, resBytes + resBytes / 2, hipHostMallocDefault));
        d->res_cap = resBytes + resBytes / 2;
    }
    auto collect = [&](bool withCells) -> int {
        hipLaunchKernelGGL(twl::collect_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, (int)n_pairs, (const int16_t *)d_err, (const int32_t *)d_alnlen,
                           (const unsigned long long *)d->cells.p, (ranMt || redoMt) ? (const unsigned long long *)d->mt_stat.p : nullptr, (unsigned long long *)d->res_h);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        const unsigned long long *r = (const unsigned long long *)d->res_h;
        const int32_t *rl = (const int32_t *)(r + 4 + n_pairs);
        const int16_t *re = (const int16_t *)(rl + n_pairs);
        std::copy(re, re + n_pairs, h_err.begin());
        d->last_alnlen.assign(rl, rl + n_pairs);
        if (withCells) { std::copy(r + 4, r + 4 + n_pairs, cells.begin()); if (ranMt || redoMt) for (int t = 0; t < 4; ++t) mtStat[t] = r[t]; }
        return TWL_OK;
    };
    bool reran = false;
    float ms_redo = 0.f;
    for (int stage = 1; stage <= 2; ++stage) {
        if ((rc = collect(!reran))) return rc;
        if (stage == 1) TRACE("dp kernel done");
        // first the pairs with an operand outside the fast division's range (lean kernels only): the IEEE-division kernel of the same window
        std::vector<int32_t> redo;
        for (int32_t n = 0; n < n_pairs; ++n) if (h_err[n] == twl::kErrGuard) redo.push_back(n);
        const bool guardRound = !redo.empty();
        if (guardRound && usedCorridor) d->corridor_lost = true;
        if (guardRound) --stage;      // (the window stages follow once these are done)
        else for (int32_t n = 0; n < n_pairs; ++n) if (h_err[n] == twl::kErrOverflow) redo.push_back(n);
        if (redo.empty()) break;
        const bool mid = (stage == 1) && (!prot || protSmall) && !startedWide;      // (a call that started on the 3072-row geometry goes on to the widest kernel)
        HIP_TRY(hipMemcpyAsync(d->items.p, redo.data(), redo.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipEventRecord(d->ev[3], st));
        int grid2 = 0, w2 = 0;
        const bool from512 = thr512 && !guardRound;      // the 512-row throughput window was outgrown: the 768-row throughput geometry takes these pairs, then the stages below
        const bool from768 = thr768 && !guardRound && !from512;      // the throughput launch's 768-row window was outgrown: first the 1024-row one (8 waves x 2 blocks), then the stages below
        if (from512) {
            thr512 = false; --stage;
            from512Pairs = (int)redo.size();
            // a few LONG pairs (8+ tiles each): all their tiles at once (1024-row stitch window) instead of one pair after the other for a pair's full latency
            long long redoLen = 0;
            for (int32_t n : redo) redoLen += (long long)h_len[2 * n] + h_len[2 * n + 1];
            const bool viaMt = (int)redo.size() <= g_mt_max_pairs && p->marker >= g_mt_min_marker && redoLen >= 8ll * p->marker * (long long)redo.size() && !d->dump_on;
            if (viaMt) {
                rc = (statMode == 5) ? launch_mt<6, 5, 3, false, 4>(d, st, a, (const int32_t *)d->items.p, redo, (int)redo.size(), h_len, &grid2, &w2)
                                     : launch_mt<6, 2, 3, false, 4>(d, st, a, (const int32_t *)d->items.p, redo, (int)redo.size(), h_len, &grid2, &w2);
                redoMt = true;       // (what outgrows that goes on to the 3072-row stage)
            } else {
                thr768 = true;
                rc = (statMode == 5) ? launch_lean<6, 4, 3, 5, 4>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2)
                                     : launch_lean<6, 4, 3, 2, 4>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2);
            }
        }
        else if (from768) {
            thr768 = false; --stage;
            rc = (statMode == 5) ? launch_lean<6, 8, 2, 5, 4>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2)
                                 : launch_lean<6, 8, 2, 2, 4>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2);
        }
        else if (guardRound) rc = prot ? launch_dp<22, 8, 2, false, true, true, 1, 3>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), 0, &grid2, &w2)
                                  : launch_dp<6, 16, 2, false, true, true, 1, 0>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), 0, &grid2, &w2);
        else if (mid && prot) rc = launch_lean<22, 16, 1, 3, 1>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2);
        // nucleotide, default matrix structure: every tile of these pairs at once on a 3072-row window (launch_mt, WIDE) when they have tiles
        // to spread; otherwise the lean kernel on a 2048-row window (8 waves x 4 blocks, reference ring still in LDS), tile after tile
        else if (mid && leanMid) {
            long long redoLen = 0;
            for (int32_t n : redo) redoLen += (long long)h_len[2 * n] + h_len[2 * n + 1];
            const bool wideMt = g_mt_wide && (int)redo.size() <= g_mt_max_pairs && p->marker >= g_mt_min_marker && redoLen >= 3ll * p->marker * (long long)redo.size();
            widePairs += (int)redo.size();
            if (wideMt) { rc = launch_mt<6, 2, 3, true, 4>(d, st, a, (const int32_t *)d->items.p, redo, (int)redo.size(), h_len, &grid2, &w2); redoMt = true; }
            else rc = launch_lean<6, 8, 4, 2, 2>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2);
        }
        else if (mid) rc = launch_dp<6, 16, 2, false, true, true, 1, 0>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), 0, &grid2, &w2);
        else rc = launch_wide((const int32_t *)d->items.p, (int)redo.size(), &grid2, &w2);
        if (rc) return rc;
        reran = true;
        HIP_TRY(hipEventRecord(d->ev[4], st));
        HIP_TRY(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, d->ev[3], d->ev[4]));
        ms_redo += ms;
        d->stats.n_launches += 1;
        d->stats.n_relaunched += (int32_t)redo.size();
        if (!mid && !guardRound && !from768 && !from512) { stage = 2; }
    }
    // how the fast window fared (see wideFirst): every pair of a small call outgrew it / the call started wide -> the streak goes on
    if (!prot && leanMid && n_run > 0 && n_run <= 8) d->wide_streak = (startedWide || (redoMt && d->stats.n_relaunched >= n_run)) ? d->wide_streak + 1 : 0;
    if (!prot && leanMid && n_run > 0) {
        if (startedWide) d->wide_calls += 1;
        else { d->last_wide_pct = (int)(100ll * widePairs / n_run); d->wide_calls = 0; }
    }
    // tiles that outgrew the 512-row window were computed in line by the stitch launch, one after the other per pair -- the expensive way to lose (10 000 x 10 kbp:
    // its levels 6 and 7, where tiles begin to outgrow 512 rows, took 34 and 33 ms instead of 23 and 16 with a 2 % allowance): ANY tile in line takes the pass off it
    if (smallTiles && g_thr_small == 0 && mtStat[1] > 0ull) d->small_state = -1;
    // how the 512-row throughput window fared (plan_nucleotide, small): a level that sent more than 1 % of its pairs on (5 % when they are long) keeps the rest
    // of the pass off it, one that fitted lets the next level start on it
    if (probed || g_thr_small != 0) {}
    else if (thr512 || from512Pairs >= 0) {
        const long long longestRun = n_run > 0 ? (long long)h_len[2 * order[0]] + h_len[2 * order[0] + 1] : 0;      // (order[0] is the longest pair, or one of the sample's: lengths of a level are alike)
        d->small_state = (std::max(from512Pairs, 0) * 100ll > (longestRun <= 4096 ? 1ll : 5ll) * n_run) ? -1 : 1;   // (long pairs re-run tile-parallel: the bet is lost later)
    }
    // a band that outgrew even the widest window (only possible with flen > 4096, i.e. in a retry of the deferred pass)
    if (reran) {
        const unsigned long long keep[4] = {mtStat[0], mtStat[1], mtStat[2], mtStat[3]};
        if ((rc = collect(true))) return rc;
        for (int t = 0; t < 4; ++t) mtStat[t] = keep[t] + (redoMt ? mtStat[t] : 0ull);      // (the counters of the first launch, plus those of a tile-parallel re-run)
    }
    // A band that outgrew even the 4608-row window (only possible with flen > 4608, i.e. in a retry of the deferred pass, alignment-cpu.cpp:116-129), or an
    // operand outside the fast division's range that met such a band: the global-memory kernel, which has no window and divides the IEEE way (round 5;
    // TWL_ERR_UNSUPPORTED ended the run here before)
    {
        std::vector<int32_t> redo;
        for (int32_t n = 0; n < n_pairs; ++n) if (h_err[n] == twl::kErrOverflow || h_err[n] == twl::kErrGuard) redo.push_back(n);
        if (!redo.empty()) {
            HIP_TRY(hipMemcpyAsync(d->items.p, redo.data(), redo.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipEventRecord(d->ev[3], st));
            int grid2 = 0, w2 = 0;
            rc = prot ? launch_global<22>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), seq_len, &grid2, &w2)
                      : launch_global<6>(d, st, a, (const int32_t *)d->items.p, (int)redo.size(), seq_len, &grid2, &w2);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(d->ev[4], st));
            HIP_TRY(hipStreamSynchronize(st));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, d->ev[3], d->ev[4]));
            ms_redo += ms;
            d->stats.n_launches += 1;
            d->stats.n_relaunched += (int32_t)redo.size();
            d->stats.window = std::max(d->stats.window, w2);
            const unsigned long long keep[4] = {mtStat[0], mtStat[1], mtStat[2], mtStat[3]};
            if ((rc = collect(true))) return rc;
            for (int t = 0; t < 4; ++t) mtStat[t] = keep[t];
        }
    }
    for (int32_t n = 0; n < n_pairs; ++n)
        if (h_err[n] == twl::kErrOverflow || h_err[n] == twl::kErrGuard) { g_err = "internal: a re-run code survived the global-memory kernel"; return TWL_ERR_HIP; }
    uint64_t total = 0;
    for (int32_t n = 0; n < n_pairs; ++n) { d->pair_cells[n] = cells[n]; total += cells[n]; }
    if (ranMt || redoMt) { d->stats.mt_tiles_predicted = (int32_t)mtStat[0]; d->stats.mt_tiles_inline = (int32_t)mtStat[1]; d->stats.mt_scouts_failed = (int32_t)mtStat[2]; }
    d->last_err = h_err;      // (twl_level_align hands them to its caller without another copy)
    // the global-memory kernel's scratch does not stay next to a resident store for the rest of the run (every launch of the call has been waited for)
    if (d->gtb.cap > ((size_t)256 << 20)) d->gtb.release();
    if (want_dbg) {
        d->dbg_host.resize((size_t)n_pairs * 16);
        HIP_TRY(hipMemcpy(d->dbg_host.data(), d->dbg.p, d->dbg_host.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int32_t n = 0; n < n_pairs && n < 64; ++n) {
            const int32_t *g = &d->dbg_host[(size_t)n * 16];
            fprintf(stderr, "[twl dbg] pair %d: tiles %d last_k %d conv 0x%x L %d U %d ref_idx %d qry_idx %d pos %d err %d steps_left %d R %d Q %d\n",
                    n, g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11]);
        }
    }
    // work done by the abandoned fast-window attempts is real GPU work but not algorithmic cells: not counted
    float ms_pack = 0.f, ms_k = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms_pack, d->ev[0], d->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms_k, d->ev[1], d->ev[2]));
    d->stats.band_cells = total;
    d->stats.nominal_cells = nominal;
    d->stats.pack_ms = ms_pack;
    d->stats.kernel_ms = ms_k + ms_redo;
    d->stats.total_ms = ms_pack + ms_k + ms_redo;
    return TWL_OK;
}

}  // namespace

static void twl_level_pool_release(Device *d);      // twl_level.inc.hip: the level buffers the device lent to its stores
namespace { extern int g_fail_next_row_allocs; void comm_destroy_raw(void *comm); }

extern "C" {

const char *twl_last_error(void) { return g_err.c_str(); }
#ifndef TWL_SOURCE_HASH
#define TWL_SOURCE_HASH "unstamped"
#endif
#ifndef TWL_BUILD_STAMP
#define TWL_BUILD_STAMP "unstamped"
#endif
// digest of every source, header and flag of this build: __graft_entry__.build() looks for it in the file and rebuilds when it is another
__attribute__((used)) static const char twl_build_stamp[] = "TWLSTAMP:" TWL_BUILD_STAMP ";";
const char *twl_version(void) { return "twilight_amd 0.5 (gfx950) src " TWL_SOURCE_HASH; }      // (the hash of the kernel sources: __graft_entry__.source_hash)

int twl_init(const int *device_ids, int n_devices)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) return TWL_OK;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count < 1) { g_err = "no HIP device"; return TWL_ERR_HIP; }
    std::vector<int> ids;
    if (!device_ids || n_devices <= 0) ids.push_back(0);
    else ids.assign(device_ids, device_ids + n_devices);
    std::vector<Device *> devs;      // committed to g_devs only when every device came up
    auto fail = [&](int rc) {
        for (auto *d : devs) {
            for (auto &e : d->ev) if (e) (void)hipEventDestroy(e);
            if (d->stream) (void)hipStreamDestroy(d->stream);
            if (d->stream2) (void)hipStreamDestroy(d->stream2);
            for (auto &e : d->ev2) if (e) (void)hipEventDestroy(e);
            delete d;
        }
        return rc;
    };
    for (int id : ids) {
        if (id < 0 || id >= count) { g_err = "device id out of range"; return fail(TWL_ERR_BAD_ARGUMENT); }
        auto *d = new Device();
        d->id = id;
        devs.push_back(d);
        hipDeviceProp_t prop;
        hipError_t e = hipSetDevice(id);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, id);
        if (e == hipSuccess) { d->num_cu = prop.multiProcessorCount; e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking); if (e == hipSuccess) { int least = 0, greatest = 0; (void)hipDeviceGetStreamPriorityRange(&least, &greatest); e = hipStreamCreateWithPriority(&d->stream2, hipStreamNonBlocking, least); } }      // (the row rewrites there give way to the DP kernels of the first stream)
        for (auto &ev : d->ev2) if (e == hipSuccess) e = hipEventCreate(&ev);
        for (auto &ev : d->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
        if (e != hipSuccess) { g_err = std::string("twl_init: ") + hipGetErrorString(e); return fail(TWL_ERR_HIP); }
    }
    g_devs = devs;
    g_init = true;
#ifdef TWL_DEV      // development builds (__graft_entry__.build() with TWL_DEV_BUILD=1): the knobs of twl_set_knob from the environment, through its clamps
    static const struct { const char *env; int knob; } kEnvKnobs[] = {{"TWL_MT_MAX_PAIRS", TWL_KNOB_MT_MAX_PAIRS}, {"TWL_MT_LEAD", TWL_KNOB_MT_LEAD}, {"TWL_MT_MARGIN", TWL_KNOB_MT_MARGIN},
        {"TWL_MT_PERTURB", TWL_KNOB_MT_PERTURB}, {"TWL_MT_ROUNDS", TWL_KNOB_MT_ROUNDS}, {"TWL_MT_THR_JOBS", TWL_KNOB_MT_THR_JOBS}, {"TWL_MT_TAIL_PCT", TWL_KNOB_MT_TAIL_PCT}};
    for (const auto &k : kEnvKnobs) if (const char *v = getenv(k.env)) (void)twl_set_knob(k.knob, atoi(v));
#endif
    return TWL_OK;
}

void twl_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    // stores (twl_level.h) point into their device's state and pool: with any of them alive the library stays up (destroy them first)
    for (auto *d : g_devs) {
        std::lock_guard<std::mutex> dl(d->mu);
        if (d->live_stores > 0) {
            fprintf(stderr, "twl_shutdown: %d store(s) still alive on device %d; the library stays initialised (twl_store_destroy them first)\n", d->live_stores, d->id);
            return;
        }
    }
    for (auto *d : g_devs) {
        (void)hipSetDevice(d->id);
        (void)hipStreamSynchronize(d->stream);
        (void)hipStreamSynchronize(d->stream2);
        twl_level_pool_release(d);
        if (d->comm) { comm_destroy_raw(d->comm); d->comm = nullptr; }
        d->comm_send.release(); d->comm_recv.release();
        for (Buf *b : {&d->cols, &d->tb, &d->gtb, &d->cells, &d->queue, &d->items, &d->errs, &d->dbg, &d->sim, &d->sim_off, &d->blk_off, &d->m24, &d->team, &d->simdump, &d->mt_chain, &d->mt_rec, &d->mt_seg, &d->mt_spath, &d->mt_stat, &d->mt_jobs, &d->mt_anchor, &d->gc_zero,
                       &d->h2d_freq, &d->h2d_gop, &d->h2d_gex, &d->h2d_len, &d->h2d_num, &d->d_aln, &d->d_alnlen, &d->d_err})
            b->release();
        for (auto &e : d->ev) if (e) (void)hipEventDestroy(e);
        if (d->stream) (void)hipStreamDestroy(d->stream);
        if (d->stream2) (void)hipStreamDestroy(d->stream2);
        for (auto &e : d->ev2) if (e) (void)hipEventDestroy(e);
        if (d->res_h) (void)hipHostFree(d->res_h);
        if (d->probe_h) (void)hipHostFree(d->probe_h);
        delete d;
    }
    g_devs.clear();
    g_init = false;
}

int twl_align_batch_device(int device, void *stream, const twl_params *p, int32_t n_pairs, int32_t seq_len, const float *d_freq,
                           const float *d_gap_open, const float *d_gap_extend, const int32_t *d_len, const int32_t *d_num,
                           int8_t *d_aln_out, int32_t *d_aln_len_out, int16_t *d_err_out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    int rc = check_params(p);
    if (rc) return rc;
    if (n_pairs < 0 || seq_len < 1) { g_err = "bad n_pairs/seq_len"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = nullptr;
    if ((rc = find_dev(device, &d))) return rc;
    std::lock_guard<std::mutex> lk(d->mu);
    hipStream_t st = stream ? (hipStream_t)stream : d->stream;
    return run_device(d, st, p, n_pairs, seq_len, d_freq, d_gap_open, d_gap_extend, d_len, d_num, d_aln_out, d_aln_len_out,
                      d_err_out, nullptr);
}

static int run_host_slice(Device *d, const twl_params *p, const std::vector<int32_t> &ids, int32_t seq_len, const float *freq,
                          const float *gop, const float *gex, const int32_t *len, const int32_t *num, int8_t *aln_out,
                          int32_t *aln_len_out, int16_t *err_out)
{
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    const int32_t n = (int32_t)ids.size();
    d->stats = twl_stats{};          // a device that gets no pairs of this call must not report the previous call's counters
    d->pair_cells.clear();
    if (n == 0) return TWL_OK;
    const size_t P = (size_t)p->P, sl = (size_t)seq_len;
    int rc;
    if ((rc = d->h2d_freq.ensure((size_t)n * 2 * sl * P * sizeof(float)))) return rc;
    if ((rc = d->h2d_gop.ensure((size_t)n * 2 * sl * sizeof(float)))) return rc;
    if ((rc = d->h2d_gex.ensure((size_t)n * 2 * sl * sizeof(float)))) return rc;
    if ((rc = d->h2d_len.ensure((size_t)n * 2 * sizeof(int32_t)))) return rc;
    if ((rc = d->h2d_num.ensure((size_t)n * 2 * sizeof(int32_t)))) return rc;
    if ((rc = d->d_aln.ensure((size_t)n * 2 * sl))) return rc;
    if ((rc = d->d_alnlen.ensure((size_t)n * sizeof(int32_t)))) return rc;
    if ((rc = d->d_err.ensure((size_t)n * sizeof(int16_t)))) return rc;
    hipStream_t st = d->stream;
    std::vector<int32_t> hl((size_t)n * 2), hn((size_t)n * 2);
    HIP_TRY(hipEventRecord(d->ev[5], st));
    bool contiguous = true;
    for (int32_t t = 0; t < n; ++t) contiguous = contiguous && ids[t] == ids[0] + t;
    for (int32_t t = 0; t < n; ++t) {
        const size_t s = (size_t)ids[t];
        hl[2 * t] = len[2 * s]; hl[2 * t + 1] = len[2 * s + 1];
        hn[2 * t] = num[2 * s]; hn[2 * t + 1] = num[2 * s + 1];
    }
    if (contiguous) {      // one transfer per array instead of three per pair
        const size_t s0 = (size_t)ids[0];
        HIP_TRY(hipMemcpyAsync(d->h2d_freq.p, freq + s0 * 2 * sl * P, (size_t)n * 2 * sl * P * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d->h2d_gop.p, gop + s0 * 2 * sl, (size_t)n * 2 * sl * sizeof(float), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d->h2d_gex.p, gex + s0 * 2 * sl, (size_t)n * 2 * sl * sizeof(float), hipMemcpyHostToDevice, st));
    } else {
        for (int32_t t = 0; t < n; ++t) {
            const size_t s = (size_t)ids[t];
            HIP_TRY(hipMemcpyAsync((float *)d->h2d_freq.p + (size_t)t * 2 * sl * P, freq + s * 2 * sl * P, 2 * sl * P * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync((float *)d->h2d_gop.p + (size_t)t * 2 * sl, gop + s * 2 * sl, 2 * sl * sizeof(float), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync((float *)d->h2d_gex.p + (size_t)t * 2 * sl, gex + s * 2 * sl, 2 * sl * sizeof(float), hipMemcpyHostToDevice, st));
        }
    }
    HIP_TRY(hipMemcpyAsync(d->h2d_len.p, hl.data(), hl.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->h2d_num.p, hn.data(), hn.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    rc = run_device(d, st, p, n, seq_len, (const float *)d->h2d_freq.p, (const float *)d->h2d_gop.p, (const float *)d->h2d_gex.p,
                    (const int32_t *)d->h2d_len.p, (const int32_t *)d->h2d_num.p, (int8_t *)d->d_aln.p, (int32_t *)d->d_alnlen.p,
                    (int16_t *)d->d_err.p, hl.data());
    if (rc) return rc;
    std::vector<int32_t> alen((size_t)n);
    std::vector<int16_t> aerr((size_t)n);
    HIP_TRY(hipMemcpyAsync(alen.data(), d->d_alnlen.p, alen.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(aerr.data(), d->d_err.p, aerr.size() * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    size_t path_bytes = 0;
    for (int32_t t = 0; t < n; ++t) path_bytes += (size_t)std::max(alen[t], 0);
    // paths: one transfer of the whole [n][2*seq_len] block when the slice is contiguous and reasonably full, else one per pair
    const bool bulk = contiguous && path_bytes * 8 >= (size_t)n * 2 * sl;
    if (bulk) HIP_TRY(hipMemcpyAsync(aln_out + (size_t)ids[0] * 2 * sl, d->d_aln.p, (size_t)n * 2 * sl, hipMemcpyDeviceToHost, st));
    for (int32_t t = 0; t < n; ++t) {
        const size_t s = (size_t)ids[t];
        aln_len_out[s] = alen[t];
        err_out[s] = aerr[t];
        if (!bulk && alen[t] > 0)
            HIP_TRY(hipMemcpyAsync(aln_out + s * 2 * sl, (int8_t *)d->d_aln.p + (size_t)t * 2 * sl, (size_t)alen[t], hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(d->ev[3], st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[5], d->ev[3]));
    d->stats.total_ms = ms;
    return TWL_OK;
}

int twl_align_batch(const twl_params *p, int32_t n_pairs, int32_t seq_len, const float *freq, const float *gap_open,
                    const float *gap_extend, const int32_t *len, const int32_t *num, int8_t *aln_out, int32_t *aln_len_out,
                    int16_t *err_out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    int rc = check_params(p);
    if (rc) return rc;
    if (n_pairs < 0 || seq_len < 1 || (n_pairs > 0 && (!freq || !gap_open || !gap_extend || !len || !num || !aln_out || !aln_len_out || !err_out))) {
        g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT;
    }
    const size_t nd = g_devs.size();
    // deal pairs to devices in descending cost order (independent units, no collective)
    std::vector<int32_t> order((size_t)n_pairs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        return (int64_t)len[2 * x] + len[2 * x + 1] > (int64_t)len[2 * y] + len[2 * y + 1];
    });
    std::vector<std::vector<int32_t>> slices(nd);
    if (nd == 1) { slices[0].resize((size_t)n_pairs); std::iota(slices[0].begin(), slices[0].end(), 0); }   // bulk copies; the device orders by cost itself
    else for (size_t t = 0; t < order.size(); ++t) slices[t % nd].push_back(order[t]);
    if (nd == 1) return run_host_slice(g_devs[0], p, slices[0], seq_len, freq, gap_open, gap_extend, len, num, aln_out, aln_len_out, err_out);
    std::vector<int> rcs(nd, 0);
    std::vector<std::string> errs(nd);
    std::vector<std::thread> th;
    for (size_t i = 0; i < nd; ++i)
        th.emplace_back([&, i] {
            rcs[i] = run_host_slice(g_devs[i], p, slices[i], seq_len, freq, gap_open, gap_extend, len, num, aln_out, aln_len_out, err_out);
            errs[i] = g_err;
        });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < nd; ++i) if (rcs[i]) { g_err = errs[i]; return rcs[i]; }
    return TWL_OK;
}

int twl_get_stats(int device, twl_stats *out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    if (!out) { g_err = "out is null"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    *out = d->stats;
    return TWL_OK;
}

// Diagnostics: similarScore(i, j) (TALCO-XDrop.cpp:444) of one pair for every (query row i, reference column j), row-major
// out[Q][R], computed by score_matrix_kernel -- the arithmetic the DP kernels use (for nucleotides the general 5x5 order).
int twl_column_scores(const twl_params *p, int32_t seq_len, const float *freq, const int32_t *len, const int32_t *num, float *out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    int rc = check_params(p);
    if (rc) return rc;
    if (seq_len < 1 || !freq || !len || !num || !out || len[0] < 1 || len[1] < 1 || len[0] > seq_len || len[1] > seq_len) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = g_devs[0];
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    hipStream_t st = d->stream;
    const size_t P = (size_t)p->P, sl = (size_t)seq_len, CW = P + 2;
    const int R = len[0], Q = len[1];
    const size_t pitch = ((size_t)Q + 63) & ~(size_t)63, simFloats = (size_t)(R + Q) * pitch;
    if ((rc = d->h2d_freq.ensure(2 * sl * P * sizeof(float)))) return rc;
    if ((rc = d->h2d_gop.ensure(2 * sl * sizeof(float)))) return rc;
    if ((rc = d->cols.ensure(2 * sl * CW * sizeof(float)))) return rc;
    if ((rc = d->h2d_len.ensure(2 * sizeof(int32_t)))) return rc;
    if ((rc = d->h2d_num.ensure(2 * sizeof(int32_t)))) return rc;
    if ((rc = d->items.ensure(sizeof(int32_t)))) return rc;
    if ((rc = d->sim.ensure(simFloats * sizeof(float)))) return rc;
    if ((rc = d->sim_off.ensure(sizeof(long long)))) return rc;
    if ((rc = d->blk_off.ensure(2 * sizeof(int32_t)))) return rc;
    if ((rc = d->m24.ensure(21 * 24 * sizeof(float)))) return rc;
    std::vector<float> m24(21 * 24, 0.0f);
    const int ms = p->P - 1;
    for (int l = 0; l < ms; ++l) for (int m = 0; m < ms; ++m) m24[(ms == 21 ? 24 : 5) * l + m] = p->matrix[ms * l + m];
    const int32_t item = 0, blk[2] = {0, (int32_t)(((R + Q - 1 + 63) / 64) * ((Q + 63) / 64))};
    const long long off = 0;
    HIP_TRY(hipMemcpyAsync(d->h2d_freq.p, freq, 2 * sl * P * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d->h2d_gop.p, 0, 2 * sl * sizeof(float), st));
    HIP_TRY(hipMemcpyAsync(d->h2d_len.p, len, 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->h2d_num.p, num, 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->items.p, &item, sizeof item, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->sim_off.p, &off, sizeof off, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->blk_off.p, blk, sizeof blk, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d->m24.p, m24.data(), m24.size() * sizeof(float), hipMemcpyHostToDevice, st));
    const size_t n_cols = 2 * sl;
    const int blocks = (int)((n_cols + 255) / 256);
    if (p->P == 22) hipLaunchKernelGGL(twl::pack_kernel<22>, dim3(blocks), dim3(256), 0, st, (const float *)d->h2d_freq.p, (const float *)d->h2d_gop.p, (const float *)d->h2d_gop.p, (float *)d->cols.p, n_cols);
    else hipLaunchKernelGGL(twl::pack_kernel<6>, dim3(blocks), dim3(256), 0, st, (const float *)d->h2d_freq.p, (const float *)d->h2d_gop.p, (const float *)d->h2d_gop.p, (float *)d->cols.p, n_cols);
    twl::ScoreArgs sa{};
    sa.cols = (const float *)d->cols.p; sa.len = (const int32_t *)d->h2d_len.p; sa.num = (const int32_t *)d->h2d_num.p;
    sa.items = (const int32_t *)d->items.p; sa.blk_off = (const int32_t *)d->blk_off.p; sa.n_items = 1; sa.seq_len = seq_len;
    sa.gap_char = p->gap_char; sa.M24 = (const float *)d->m24.p; sa.sim = (float *)d->sim.p; sa.sim_off = (const long long *)d->sim_off.p;
    if (p->P == 22) hipLaunchKernelGGL(twl::score_matrix_kernel<22>, dim3((unsigned)blk[1]), dim3(256), 0, st, sa);
    else hipLaunchKernelGGL(twl::score_matrix_kernel<6>, dim3((unsigned)blk[1]), dim3(256), 0, st, sa);
    HIP_TRY(hipGetLastError());
    std::vector<float> diag(simFloats);
    HIP_TRY(hipMemcpyAsync(diag.data(), d->sim.p, simFloats * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < Q; ++i)
        for (int j = 0; j < R; ++j) out[(size_t)i * R + j] = diag[(size_t)(i + j) * pitch + i];
    return TWL_OK;
}

int twl_dp_column_scores(const twl_params *p, int32_t seq_len, const float *freq, const float *gap_open, const float *gap_extend,
                         const int32_t *len, const int32_t *num, float *out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    int rc = check_params(p);
    if (rc) return rc;
    if (seq_len < 1 || !freq || !gap_open || !gap_extend || !len || !num || !out || len[0] < 1 || len[1] < 1 || len[0] > seq_len || len[1] > seq_len) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = g_devs[0];
    const size_t cells = (size_t)len[0] * (size_t)len[1];
    {
        std::lock_guard<std::mutex> lk(d->mu);
        HIP_TRY(hipSetDevice(d->id));
        if ((rc = d->simdump.ensure(cells * sizeof(float)))) return rc;
        HIP_TRY(hipMemset(d->simdump.p, 0xff, cells * sizeof(float)));      // NaN: a cell the band never visited
        d->dump_on = true;
    }
    std::vector<int8_t> aln(2 * (size_t)seq_len);
    int32_t alen = 0;
    int16_t aerr = 0;
    rc = twl_align_batch(p, 1, seq_len, freq, gap_open, gap_extend, len, num, aln.data(), &alen, &aerr);
    std::lock_guard<std::mutex> lk(d->mu);
    d->dump_on = false;
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipMemcpy(out, d->simdump.p, cells * sizeof(float), hipMemcpyDeviceToHost));
    return TWL_OK;
}

// debug aid (not in the public header): first `n` 64-bit words of the last call's debug record (TWL_DEBUG=1)
int twl_debug_read(int device, long long *out, int32_t n)
{
    Device *d = nullptr;
    if (find_dev(device, &d)) return TWL_ERR_BAD_ARGUMENT;
    // the stamp record (TWL_KERNEL_STAMPS builds) sits after the per-pair records
    const size_t base = d->pair_cells.size() * 16 * sizeof(int32_t);
    if (!d->dbg.p || base + (size_t)n * 8 > d->dbg.cap) return TWL_ERR_BAD_ARGUMENT;
    return hipMemcpy(out, (const char *)d->dbg.p + base, (size_t)n * 8, hipMemcpyDeviceToHost) == hipSuccess ? TWL_OK : TWL_ERR_HIP;
}

void *twl_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (!g_init || bytes == 0) return nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void twl_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int twl_copy_to_device(int device, void *dst_dev, const void *src, uint64_t bytes)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    if (bytes == 0) return TWL_OK;
    if (!dst_dev || !src) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipMemcpy(dst_dev, src, (size_t)bytes, hipMemcpyHostToDevice));
    return TWL_OK;
}

int twl_copy_from_device(int device, void *dst, const void *src_dev, uint64_t bytes)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    if (bytes == 0) return TWL_OK;
    if (!dst || !src_dev) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipMemcpy(dst, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_copy_rows_from_device(int device, void *dst, uint64_t dst_pitch, const void *src_dev, uint64_t src_pitch, uint64_t width, uint64_t rows)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    if (width == 0 || rows == 0) return TWL_OK;
    if (!dst || !src_dev || dst_pitch < width || src_pitch < width) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    HIP_TRY(hipSetDevice(d->id));
    HIP_TRY(hipMemcpy2D(dst, (size_t)dst_pitch, src_dev, (size_t)src_pitch, (size_t)width, (size_t)rows, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_set_knob(int key, int value)
{
    std::lock_guard<std::mutex> lk(g_mu);
    switch (key) {
    case TWL_KNOB_MT_PERTURB: g_mt_perturb = std::max(0, value); return TWL_OK;
    case TWL_KNOB_MT_MAX_PAIRS: g_mt_max_pairs = std::max(0, value); return TWL_OK;
    case TWL_KNOB_MT_MIN_MARKER: g_mt_min_marker = std::max(2, value); return TWL_OK;
    case TWL_KNOB_MT_LEAD: g_mt_lead = std::max(16, value); return TWL_OK;
    case TWL_KNOB_MT_MARGIN: if (value < 0) { g_mt_marg = 40; g_mt_marg_lat = 64; } else g_mt_marg = g_mt_marg_lat = std::max(2, value); return TWL_OK;      // (negative: the defaults of both kinds of level)
    case TWL_KNOB_MT_ROUNDS: g_mt_rounds = std::max(1, std::min(7, value)); return TWL_OK;
    case TWL_KNOB_MT_THR_JOBS: g_mt_thr_jobs = std::max(0, value); return TWL_OK;
    case TWL_KNOB_FAIL_ROW_ALLOCS: g_fail_next_row_allocs = std::max(0, value); return TWL_OK;
    case TWL_KNOB_PROT_MODE: if (value < 0 || value > 6) { g_err = "protein mode 0..6"; return TWL_ERR_BAD_ARGUMENT; } g_prot_mode = value; return TWL_OK;
    case TWL_KNOB_ASSUME_ONEHOT_QUERY: g_assume_onehot_query = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_MT_TAIL_PCT: g_mt_tail_pct = std::max(0, std::min(100, value)); return TWL_OK;
    case TWL_KNOB_MT_WIDE: g_mt_wide = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_NO_SPEC: g_no_spec = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_FORCE_GLOBAL: g_force_global = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_LEAF_STEP: g_leaf_step = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_POISON_TB: g_poison_tb = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_MT_ANCHOR: g_mt_anchor = value ? 1 : 0; return TWL_OK;
    case TWL_KNOB_MT_LEAD2: if (value < 0) { g_mt_lead2 = 96; g_mt_lead2_lat = 128; } else g_mt_lead2 = g_mt_lead2_lat = std::max(16, value); return TWL_OK;
    case TWL_KNOB_PROT_CORRIDOR: g_prot_corridor = std::max(0, value); for (auto *d : g_devs) { d->corridor_lost = false; d->corridor_last_n = 0; } return TWL_OK;      // (and forgets what earlier levels found)
    case TWL_KNOB_SCOUT_XDROP_PCT: g_scout_xdrop_pct = std::max(10, std::min(100, value)); return TWL_OK;
    case TWL_KNOB_THR_SMALL: g_thr_small = std::max(0, std::min(2, value)); for (auto *d : g_devs) d->small_state = d->small_last_n = 0; return TWL_OK;      // (and forgets what earlier levels found)
    default: g_err = "unknown knob"; return TWL_ERR_BAD_ARGUMENT;
    }
}

#include "twl_comm.inc.hip"

// The launch plan of a nucleotide call, as run_device would make it, in words: no device is touched (unit tests of the policy on a CPU-only box).
int twl_plan_describe(const twl_params *p, int32_t n_pairs, const int32_t *len, int32_t num_cu, int32_t qry_onehot, int32_t wide_streak, char *out, int32_t cap)
{
    // (wide_streak >= 1000 encodes the other memory of the device: 1000 + 10 * calls started wide + (1 if three quarters of the last narrow-first call went wide);
    //  + 100000 * (32 + what the pass remembers of the 512-row throughput window, NucFacts::small_state: -1 / 1) when that is not 0)
    const int small_state = wide_streak >= 100000 ? wide_streak / 100000 - 32 : 0;
    wide_streak %= 100000;
    if (!p || p->P != 6 || n_pairs < 0 || (n_pairs > 0 && !len) || num_cu < 1 || !out || cap < 64) { g_err = "bad argument (nucleotide parameters, a buffer of 64+ bytes)"; return TWL_ERR_BAD_ARGUMENT; }
    std::vector<int32_t> order;
    for (int32_t n = 0; n < n_pairs; ++n) if (len[2 * n] > 0 && len[2 * n + 1] > 0) order.push_back(n);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return (int64_t)len[2 * x] + len[2 * x + 1] > (int64_t)len[2 * y] + len[2 * y + 1]; });
    float M[25];
    for (int t = 0; t < 25; ++t) M[t] = p->matrix[t];
    NucFacts nf;
    nf.n_run = (int)order.size(); nf.num_cu = num_cu; nf.marker = p->marker; nf.M = M; nf.gap_char = p->gap_char; nf.qry_onehot = qry_onehot != 0; nf.wide_streak = wide_streak < 1000 ? wide_streak : 0;
    if (wide_streak >= 1000) { nf.last_wide_pct = ((wide_streak - 1000) % 10) ? 100 : 0; nf.wide_calls = (wide_streak - 1000) / 10; }
    nf.small_state = small_state;
    nf.h_len = len; nf.order = order.data();
    const NucPlan pl = plan_nucleotide(nf, current_knobs());
    snprintf(out, (size_t)cap, "%s; mode %d; window %d%s; bulk %d tail %d", nuc_first_name(pl.first), pl.mm5 ? 5 : pl.mm,
             pl.first == NucFirst::WideMt ? 3072 : ((pl.first == NucFirst::Throughput && pl.small) ? 512 : ((pl.first == NucFirst::Throughput && pl.four) ? 768 : 1024)), pl.probe ? " or 768 (a sample of the level decides)" : "", pl.bulk, pl.tail);
    return TWL_OK;
}

int twl_get_pair_cells(int device, uint64_t *cells_out, int32_t n)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    if (!cells_out || n < 0 || (size_t)n > d->pair_cells.size()) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    for (int32_t i = 0; i < n; ++i) cells_out[i] = d->pair_cells[i];
    return TWL_OK;
}

}  // extern "C"

#include "twl_level.inc.hip"

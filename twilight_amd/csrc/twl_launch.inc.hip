// twilight_amd/csrc/twl_launch.inc.hip -- one function per kernel family that sizes the scratch, fills the kernel arguments and launches: round-1 kernels, the global-memory kernel, the lean kernels, the tile-parallel launches of a level.
// Included by twl_align.hip (one translation unit: it shares that file's Device bookkeeping, error string and fill queue).

template <class CfgT>
size_t tb_words_for(int marker) { return ((size_t)(marker >> 3) + 1) * (size_t)CfgT::WINDOW; }

template <int P, int W, int RPL, bool PRE, bool REFLDS, bool QREG = true, int MINW = 1, int MM = 0>
int launch_dp(Device *d, hipStream_t st, const twl::KArgs &base, const int32_t *d_items, int n_items, int blocks_per_cu, int *grid_out,
              int *window_out = nullptr)
{
    using CfgT = twl::Cfg<P, W, RPL, PRE, REFLDS, QREG>;
    if (blocks_per_cu <= 0) {
        static std::atomic<int> cached{0};      // one value per template instantiation (device threads may race to fill it: same value)
        if (cached.load() == 0) {
            int nb = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (twl::talco_kernel<P, W, RPL, PRE, REFLDS, QREG, MINW, MM>), CfgT::THREADS, 0));
            cached.store(std::max(1, nb));
        }
        blocks_per_cu = cached.load();
    }
    if (window_out) *window_out = CfgT::WINDOW;
    int grid = std::min(n_items, d->num_cu * std::max(1, blocks_per_cu));
    if (grid < 1) grid = 1;
    const size_t tbw = tb_words_for<CfgT>(base.marker);
    int rc = d->tb.ensure(tbw * sizeof(uint32_t) * (size_t)grid);
    if (rc) return rc;
    if (g_poison_tb) HIP_TRY(hipMemsetAsync(d->tb.p, 0xFF, tbw * sizeof(uint32_t) * (size_t)grid, st));
    twl::KArgs a = base;
    a.tb = (uint32_t *)d->tb.p;
    a.tb_words = (int32_t)tbw;
    a.items = d_items;
    a.n_items = n_items;
    FILL_TRY(queue_fill(d, st, d->queue.p, sizeof(int32_t), 0));
    int32_t *hb = nullptr;
#ifdef TWL_KERNEL_DEBUG
    if (dbg_on()) {
        HIP_TRY(hipHostMalloc((void **)&hb, 16 * sizeof(int32_t), hipHostMallocMapped));
        for (int i = 0; i < 16; ++i) hb[i] = -777;
        a.hb = hb;
    }
#endif
    TRACE("launch dp W=%d RPL=%d grid=%d threads=%d n_items=%d tb_words=%zu", W, RPL, grid, CfgT::THREADS, n_items, tbw);
    if (!d->kname[0]) snprintf(d->kname, sizeof d->kname, "talco_kernel<%d, %d, %d, %s, %s, %s, %d, %d>", P, W, RPL, PRE ? "true" : "false", REFLDS ? "true" : "false", QREG ? "true" : "false", MINW, MM);
    FILL_TRY(flush_fills(d, st));
    hipLaunchKernelGGL((twl::talco_kernel<P, W, RPL, PRE, REFLDS, QREG, MINW, MM>), dim3(grid), dim3(CfgT::THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    if (hb) {   // debug only: poll the heartbeat until the kernel is done (or 20 s)
        for (int t = 0; t < 200; ++t) {
            if (hipStreamQuery(st) == hipSuccess) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (t % 10 == 9)
                TRACE("hb: start %d item %d tile %d k %d preB %d postB %d preTB %d postTB %d n %d err %d done %d", hb[0], hb[1], hb[2], hb[3],
                      hb[4], hb[5], hb[6], hb[7], hb[8], hb[9], hb[10]);
        }
        TRACE("hb final: start %d item %d tile %d k %d preB %d postB %d preTB %d postTB %d n %d err %d done %d", hb[0], hb[1], hb[2], hb[3],
              hb[4], hb[5], hb[6], hb[7], hb[8], hb[9], hb[10]);
    }
    *grid_out = grid;
    return TWL_OK;
}

// The last stage of the re-run chain (talco_global.hip.h): DP rows in global scratch, any band width.  One workgroup of 1024 threads per pair; as many
// workgroups as the pairs need, within a scratch budget (a 30 kbp pair at marker 1024 takes ~33 MB: 14 rows of fLen + 2 words and (marker + 2) rows of pointer bytes).
template <int P>
int launch_global(Device *d, hipStream_t st, const twl::KArgs &base, const int32_t *d_items, int n_items, int32_t seq_len, int *grid_out, int *window_out)
{
    const size_t rowcap = (size_t)std::min(std::max(base.flen, 1), std::max(seq_len, 1)) + 2;
    const size_t words = 14 * rowcap + (((size_t)base.marker + 2) * rowcap + 3) / 4;
    const size_t budget = (size_t)4 << 30;
    // one pair's scratch may not be larger than the budget either: refused (the caller sees TWL_ERR_UNSUPPORTED), not silently pinned in HBM
    if (words * sizeof(uint32_t) > budget) { g_err = "a pair whose band needs more than 4 GiB of scratch in the global-memory kernel"; return TWL_ERR_UNSUPPORTED; }
    int grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_items, budget / (words * sizeof(uint32_t))));
    grid = std::min(grid, d->num_cu);
    // a buffer of its own, handed back after the stage (run_device: release_global_scratch): d->tb lives until twl_shutdown and a single retry of the deferred
    // pass would otherwise pin several GiB next to the resident store for the rest of the run
    int rc = d->gtb.ensure(words * sizeof(uint32_t) * (size_t)grid);
    if (rc) return rc;
    if (g_poison_tb) HIP_TRY(hipMemsetAsync(d->gtb.p, 0xFF, words * sizeof(uint32_t) * (size_t)grid, st));
    twl::GArgs g;
    g.k = base;
    g.k.tb = (uint32_t *)d->gtb.p;
    g.k.tb_words = (int32_t)words;
    if ((size_t)g.k.tb_words != words) { g_err = "a pair too long for the global-memory kernel's scratch index"; return TWL_ERR_UNSUPPORTED; }
    g.k.items = d_items;
    g.k.n_items = n_items;
    g.rowcap = (int32_t)rowcap;
    FILL_TRY(queue_fill(d, st, d->queue.p, sizeof(int32_t), 0));
    FILL_TRY(flush_fills(d, st));
    TRACE("launch global P=%d grid=%d n_items=%d rowcap=%zu words=%zu", P, grid, n_items, rowcap, words);
    hipLaunchKernelGGL((twl::talco_global_kernel<P>), dim3(grid), dim3(1024), 0, st, g);
    HIP_TRY(hipGetLastError());
    *grid_out = grid;
    if (window_out) *window_out = (int)rowcap - 2;
    return TWL_OK;
}

// The round-2 nucleotide kernel (talco_nuc.hip.h): same launch protocol as launch_dp.
template <int P, int W, int RPL, int MM, int MINW, bool SPEC = false, bool DUMP = false, int SP = 0>
int launch_lean(Device *d, hipStream_t st, const twl::KArgs &base, const int32_t *d_items, int n_items, int *grid_out, int *window_out)
{
    using CfgT = twl::NCfg<W, RPL>;
    static std::atomic<int> cached{0};      // one value per template instantiation
    if (cached.load() == 0) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (twl::talco_lean_kernel<P, W, RPL, MM, MINW, SPEC, DUMP, 0, SP>), CfgT::THREADS, 0));
        cached.store(std::max(1, nb));
    }
    int blocks_per_cu = cached.load();
    if (window_out) *window_out = CfgT::WINDOW;
    int grid = std::min(n_items, d->num_cu * blocks_per_cu);
    if (grid < 1) grid = 1;
    if (SPEC) {      // two workgroups per pair that wait for each other: all of them must be resident at once
        grid = 2 * n_items;
        if (grid > d->num_cu * blocks_per_cu) { g_err = "speculative launch larger than the device"; return TWL_ERR_BAD_ARGUMENT; }
    }
    const size_t tbw = ((size_t)(base.marker >> 3) + 1) * (size_t)CfgT::WINDOW;
    int rc = d->tb.ensure(tbw * sizeof(uint32_t) * (size_t)grid);
    if (rc) return rc;
    if (g_poison_tb) HIP_TRY(hipMemsetAsync(d->tb.p, 0xFF, tbw * sizeof(uint32_t) * (size_t)grid, st));
    twl::NArgs a{};
    a.cols = base.cols; a.len = base.len; a.num = base.num; a.aln = base.aln; a.aln_len = base.aln_len; a.err = base.err;
    a.cells = base.cells; a.tb = (uint32_t *)d->tb.p; a.queue = base.queue; a.items = d_items; a.n_items = n_items;
    a.seq_len = base.seq_len; a.tb_words = (int32_t)tbw; a.dbg = base.dbg; a.n_pairs_total = base.n_pairs_total;
    a.step_slack = base.step_slack; a.gap_open = base.gap_open; a.gap_extend = base.gap_extend; a.gap_char = base.gap_char; a.gc_zero = base.gc_zero;
    a.xdrop = base.xdrop; a.flen = base.flen; a.marker = base.marker;
    for (int t = 0; t < 25; ++t) a.M[t] = base.M[t];
    a.M24 = (const float *)d->m24.p; a.sim = base.sim; a.sim_off = base.sim_off;
    FILL_TRY(queue_fill(d, st, d->queue.p, sizeof(int32_t), 0));
    if (SPEC) {
        if ((rc = d->team.ensure((size_t)n_items * twl::kTeamWords * sizeof(unsigned long long)))) return rc;
        FILL_TRY(queue_fill(d, st, d->team.p, (size_t)n_items * twl::kTeamWords * sizeof(unsigned long long), 0));
        FILL_TRY(queue_fill(d, st, a.cells, (size_t)base.n_pairs_total * sizeof(unsigned long long), 0));
        a.team = (unsigned long long *)d->team.p;
    }
    TRACE("launch lean P=%d W=%d RPL=%d MM=%d grid=%d threads=%d n_items=%d tb_words=%zu", P, W, RPL, MM, grid, CfgT::THREADS, n_items, tbw);
    if (!d->kname[0]) {
        if constexpr (SP != 0) snprintf(d->kname, sizeof d->kname, "talco_lean_kernel<%d, %d, %d, %d, %d, %s, %s, 0, %d>", P, W, RPL, MM, MINW, SPEC ? "true" : "false", DUMP ? "true" : "false", SP);
        else snprintf(d->kname, sizeof d->kname, "talco_lean_kernel<%d, %d, %d, %d, %d, %s, %s, 0>", P, W, RPL, MM, MINW, SPEC ? "true" : "false", DUMP ? "true" : "false");
    }
    a.simdump = DUMP ? (float *)d->simdump.p : nullptr;
    FILL_TRY(flush_fills(d, st));
    hipLaunchKernelGGL((twl::talco_lean_kernel<P, W, RPL, MM, MINW, SPEC, DUMP, 0, SP>), dim3(grid), dim3(CfgT::THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    if (SPEC && dbg_on()) {      // development: how often the guessed tile start was the true one
        std::vector<unsigned long long> tw((size_t)n_items * twl::kTeamWords);
        HIP_TRY(hipMemcpyAsync(tw.data(), d->team.p, tw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        unsigned long long g = 0, h = 0;
        for (int t = 0; t < n_items; ++t) { g += tw[(size_t)t * twl::kTeamWords + twl::kTeamStat]; h += tw[(size_t)t * twl::kTeamWords + twl::kTeamStat + 1]; }
        fprintf(stderr, "[twl spec] %d pairs: %llu tile starts guessed, %llu confirmed\n", n_items, g, h);
    }
    *grid_out = grid;
    return TWL_OK;
}

constexpr int kMtMaxRounds = 7, kMtCounters = 16;      // 1 + 2 * rounds launches, each with its own work counter (the count is clamped wherever it is set)
// One launch of a tile-parallel kernel (MT 1 tiles / 2 scouts / 3 stitch) of geometry <W, RPL>; the caller has filled the NArgs.
template <int P, int W, int RPL, int MM, int MINW, int MT, int SP = 0>
int launch_mt_kernel(Device *d, hipStream_t st, twl::NArgs a, int n_items, int *grid_out = nullptr, size_t tb_groups = 0)
{
    using CfgT = twl::NCfg<W, RPL>;
    static std::atomic<int> cached{0};
    if (cached.load() == 0) {
        int nb = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (twl::talco_lean_kernel<P, W, RPL, MM, MINW, false, false, MT, SP>), CfgT::THREADS, 0));
        cached.store(std::max(1, nb));
    }
    int grid = std::max(1, std::min(n_items, d->num_cu * cached.load()));
    // traceback words of a workgroup: the groups of 8 anti-diagonals up to the marker, or (pair scouts, MT 4) of a whole pair
    const size_t tbw = (tb_groups ? tb_groups : (size_t)(a.marker >> 3) + 1) * (size_t)CfgT::WINDOW;
    if (tb_groups) grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)grid, ((size_t)8 << 30) / (tbw * sizeof(uint32_t))));      // (at most 8 GB of them: fewer workgroups take the pairs in turn)
    int rc = d->tb.ensure(tbw * sizeof(uint32_t) * (size_t)grid);
    if (rc) return rc;
    if (g_poison_tb) HIP_TRY(hipMemsetAsync(d->tb.p, 0xFF, tbw * sizeof(uint32_t) * (size_t)grid, st));
    a.tb = (uint32_t *)d->tb.p; a.tb_words = (int32_t)tbw; a.n_items = n_items;
    // (every launch of a tile-parallel level has its own work counter: launch_mt zeroed the 16 of them in one go)
    a.queue = (int32_t *)d->queue.p + (d->mt_launch++ % kMtCounters);
    FILL_TRY(flush_fills(d, st));
    hipLaunchKernelGGL((twl::talco_lean_kernel<P, W, RPL, MM, MINW, false, false, MT, SP>), dim3(grid), dim3(CfgT::THREADS), 0, st, a);
    HIP_TRY(hipGetLastError());
    if (grid_out) *grid_out = grid;
    return TWL_OK;
}


// What the host knows about every pair of a nucleotide launch, taken out of the kernel's per-block tests (talco_lean_kernel, SP): 1 = leaf x leaf (needs the
// one-letter-query column score, MM 5).  Only the throughput geometries have the specialised step.
template <int W, int RPL, int MINW>
int launch_thr(Device *d, hipStream_t st, const twl::KArgs &a, const int32_t *items, int n, int *grid, int *window, bool mm5, int sp)
{
    if (mm5) {
        if (sp == 1) return launch_lean<6, W, RPL, 5, MINW, false, false, 1>(d, st, a, items, n, grid, window);
        return launch_lean<6, W, RPL, 5, MINW>(d, st, a, items, n, grid, window);
    }
    return launch_lean<6, W, RPL, 2, MINW>(d, st, a, items, n, grid, window);
}
template <int W, int RPL, int MM, int MINW, int MT>
int launch_mt_thr(Device *d, hipStream_t st, const twl::NArgs &a, int n_items, int *grid_out, int sp)
{
    if constexpr (MM == 5) { if (sp == 1) return launch_mt_kernel<6, W, RPL, MM, MINW, MT, 1>(d, st, a, n_items, grid_out); }
    return launch_mt_kernel<6, W, RPL, MM, MINW, MT>(d, st, a, n_items, grid_out);
}

// Tile-parallel alignment of a level with few pairs (talco_nuc.hip.h, MT kernels): scouts, then rounds of chain -> tiles -> stitch, on `st`.
// `order` = the pairs that run, h_len their lengths on the host.  Scouts and tiles run on the 16-wave geometry (one workgroup per CU, the
// shortest diagonal step) while they fit the device at once (one workgroup per CU), on the throughput geometry (8 waves x 2 blocks, two per CU) beyond.

template <int P, int MM, int TRPL, bool WIDE = false, int TW = 8>      // TW x TRPL: waves and 64-row blocks per wave of the throughput geometry (nucleotide 4 x 3: 768 rows, four workgroups per CU; protein 8 x 1: 512 rows, two)
int launch_mt(Device *d, hipStream_t st, const twl::KArgs &base, const int32_t *d_items, const std::vector<int32_t> &order, int n_run,
              const int32_t *h_len, int *grid_out, int *window_out, bool small_tiles = false, int sp = 0)
{
    // small_tiles (nucleotide, throughput geometry): the tile jobs too run on 4 waves x 2 blocks, five workgroups per CU -- the earlier levels of this pass
    // fitted the 512-row window (plan_nucleotide, small).  A tile that outgrows it leaves a failed record of a window narrower than the stitch launch's, which
    // the stitch launch does not adopt: it computes that tile in line (run_device counts them and takes the pass off the small window when they are many).
    constexpr bool kCanSmall = (P == 6 && TW == 4 && !WIDE);
    const bool smallT = kCanSmall && small_tiles;
    // WIDE (nucleotide): the tiles and the stitch launch on 16 waves x 3 blocks, a 3072-row window -- for the pairs whose band outgrew the
    // 1024-row window of the fast geometries.  Until round 4 those ran their ~20 tiles one after the other on the 2048- and 4608-row kernels
    // (0.2-0.6 s per 10 kbp pair); their tiles are as independent as anybody's.  The scouts keep the narrow geometry: a scout's band opens
    // from one cell by a row per diagonal over the ~400 diagonals it runs.
#if defined(TWL_EXP_LAT_W)      // geometry experiments of the latency launches (tools/lone_pair_probe.py on cross-compiled variants)
    constexpr int SW = WIDE ? 16 : TWL_EXP_LAT_W, SR = WIDE ? 3 : TWL_EXP_LAT_RPL;
#else
    constexpr int SW = 16, SR = WIDE ? 3 : 1;       // geometry of the stitch launch and of tiles while they fit the device at once
#endif
    if (window_out) *window_out = twl::NCfg<SW, SR>::WINDOW;
    const int marker = base.marker;
    const int slots = (2 * base.seq_len) / (marker - 1) + 2;
    const int segcap = 2 * marker + 16;
    const int sp_pitch = 2 * base.seq_len + 8;
    const size_t np = (size_t)n_run;               // the tables are indexed by the position of a pair in `order` (not by pair id)
    // jobs {pair, slot, row}: scouts for every tile boundary t >= 1 that exists, tiles for t >= 0 (tile-major, so that the tiles of a pair spread over the launch)
    std::vector<int32_t> &jobs = d->mt_jobs_host;       // scouts first, then tiles (kept with the device: the upload below is asynchronous)
    jobs.clear();
    int maxT = 0;
    std::vector<int> T((size_t)n_run);
    for (int t = 0; t < n_run; ++t) {
        const int pr = order[t];
        const long long RQ = (long long)h_len[2 * pr] + h_len[2 * pr + 1];
        int n = 1;
        while (n < slots && (long long)(marker - 1) * n - 1 <= RQ - 2) ++n;
        T[t] = n; maxT = std::max(maxT, n);
    }
    for (int s = 1; s < maxT; ++s) for (int t = 0; t < n_run; ++t) if (s < T[t]) { jobs.push_back(order[t]); jobs.push_back(s); jobs.push_back(t); }
    const int nScout = (int)(jobs.size() / 3);
    for (int s = 0; s < maxT; ++s) for (int t = 0; t < n_run; ++t) if (s < T[t]) { jobs.push_back(order[t]); jobs.push_back(s); jobs.push_back(t); }
    const int nTile = (int)(jobs.size() / 3) - nScout;
    int rc;
    if ((rc = d->mt_chain.ensure(np * slots * 2 * sizeof(int32_t)))) return rc;
    if ((rc = d->mt_rec.ensure(np * slots * twl::kMtRec * sizeof(int32_t)))) return rc;
    if ((rc = d->mt_seg.ensure(np * slots * (size_t)segcap))) return rc;
    if ((rc = d->mt_spath.ensure(np * (size_t)sp_pitch * sizeof(int32_t)))) return rc;
    if ((rc = d->mt_stat.ensure(4 * sizeof(unsigned long long) + np * 8 * sizeof(int32_t)))) return rc;      // counters, then the per-pair frontier
    if ((rc = d->mt_jobs.ensure(jobs.size() * sizeof(int32_t)))) return rc;
    // scouts of the narrow geometries start from an anchored cell where the profiles give one (talco_nuc.hip.h, mt_anchor_kernel)
    const bool anchors = !WIDE && g_mt_anchor && nScout > 0;
    if (anchors && (rc = d->mt_anchor.ensure(np * slots * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(d->mt_jobs.p, jobs.data(), jobs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    FILL_TRY(queue_fill(d, st, d->mt_rec.p, np * slots * twl::kMtRec * sizeof(int32_t), 0));
    FILL_TRY(queue_fill(d, st, d->mt_spath.p, np * (size_t)sp_pitch * sizeof(int32_t), 0xFE));
    FILL_TRY(queue_fill(d, st, d->mt_stat.p, 4 * sizeof(unsigned long long) + np * 8 * sizeof(int32_t), 0));
    if (anchors) FILL_TRY(queue_fill(d, st, d->mt_anchor.p, np * slots * sizeof(int32_t), 0xFF));
    const int rounds = std::max(1, std::min(g_mt_rounds, kMtMaxRounds));
    FILL_TRY(queue_fill(d, st, d->queue.p, kMtCounters * sizeof(int32_t), 0));      // one work counter per launch: scouts + rounds x (tiles, stitch)
    d->mt_launch = 0;
    twl::NArgs a{};
    a.cols = base.cols; a.len = base.len; a.num = base.num; a.aln = base.aln; a.aln_len = base.aln_len; a.err = base.err;
    a.cells = base.cells; a.queue = base.queue; a.items = d_items;
    a.seq_len = base.seq_len; a.dbg = nullptr; a.n_pairs_total = base.n_pairs_total;
    a.step_slack = base.step_slack; a.gap_open = base.gap_open; a.gap_extend = base.gap_extend; a.gap_char = base.gap_char; a.gc_zero = base.gc_zero;
    a.xdrop = base.xdrop; a.flen = base.flen; a.marker = base.marker;
    for (int t = 0; t < 25; ++t) a.M[t] = base.M[t];
    a.M24 = (const float *)d->m24.p; a.sim = base.sim; a.sim_off = base.sim_off;
    a.mt_chain = (int32_t *)d->mt_chain.p; a.mt_rec = (int32_t *)d->mt_rec.p; a.mt_seg = (int8_t *)d->mt_seg.p; a.mt_spath = (int32_t *)d->mt_spath.p;
    a.mt_stat = (unsigned long long *)d->mt_stat.p;
    a.mt_front = (int32_t *)((unsigned long long *)d->mt_stat.p + 4);
    const bool thr = !WIDE && nTile > g_mt_thr_jobs;
    // (see twl_knobs.inc.hip: longer scouts where the tile jobs are at most a round or two of the device's workgroups and a missed start a second round that lasts as long as the first)
    const bool longScouts = !WIDE && nTile <= 2048;
    const int lead2 = longScouts ? g_mt_lead2_lat : g_mt_lead2;
    a.mt_slots = slots; a.mt_segcap = segcap; a.mt_sp_pitch = sp_pitch; a.mt_lead = g_mt_lead; a.mt_marg = longScouts ? g_mt_marg_lat : g_mt_marg;
    TRACE("launch mt pairs=%d scouts=%d tiles=%d slots=%d geometry=%s", n_run, nScout, nTile, slots, WIDE ? "16x3 (wide)" : (thr ? "throughput geometry" : "16x1"));
    if (!d->kname[0]) {
        if (thr && smallT) snprintf(d->kname, sizeof d->kname, "talco_lean_kernel<%d, 4, 2, %d, 5, false, false, 2 / 1> + <%d, 16, 1, %d, 1, false, false, 3> (tile-parallel: scouts, tiles, stitch)", P, MM, P, MM);
        else if (thr) snprintf(d->kname, sizeof d->kname, "talco_lean_kernel<%d, %d, %d, %d, 4, false, false, 2 / 1> + <%d, 16, 1, %d, 1, false, false, 3> (tile-parallel: scouts, tiles, stitch)", P, TW, TRPL, MM, P, MM);
        else snprintf(d->kname, sizeof d->kname, "talco_lean_kernel<%d, 16, %d, %d, 1, false, false, 2 / 1 / 3> (tile-parallel: scouts, tiles, stitch)", P, SR, MM);
    }
    if (WIDE) {
        // pairs that outgrew the fast window are, more often than not, pairs whose tiles converge late or never (diffuse profiles): a tile then runs
        // to the end of the pair and the next one starts where the path from the END cell crosses the marker diagonal -- nothing a scout that
        // starts 320 diagonals ahead can know.  One workgroup per pair runs the whole DP once with every traceback word kept (MT 4) and leaves the
        // global path's crossing of every anti-diagonal; its suffixes are what the tiles' own paths follow.
        int maxRQ = 0;
        for (int t = 0; t < n_run; ++t) maxRQ = std::max(maxRQ, h_len[2 * order[t]] + h_len[2 * order[t] + 1]);
        if constexpr (P == 6) {
            twl::NArgs as = a;
            as.xdrop = (int32_t)((long long)a.xdrop * g_scout_xdrop_pct / 100);
            if ((rc = launch_mt_kernel<P, SW, SR, MM, 1, 4>(d, st, as, n_run, nullptr, (size_t)(maxRQ >> 3) + 2))) return rc;
        }
    } else if (nScout > 0) {
        a.mt_jobs = (const int32_t *)d->mt_jobs.p;
        if (anchors) {
            FILL_TRY(flush_fills(d, st));
            hipLaunchKernelGGL(twl::mt_anchor_kernel<P>, dim3(nScout), dim3(256), 0, st, base.cols, base.len, base.seq_len, a.mt_jobs, nScout, (int32_t *)d->mt_anchor.p, slots, marker, lead2);
            HIP_TRY(hipGetLastError());
            a.mt_anchor = (const int32_t *)d->mt_anchor.p; a.mt_lead2 = lead2;
        }
        const bool thrS = WIDE ? nScout > g_mt_thr_jobs : thr;
        // (nucleotide scouts of the throughput geometry run ~330 diagonals from one cell: their band opens by a row per diagonal and cannot outgrow 449 rows, so
        //  they take the 512-row window -- 4 waves x 2 blocks, FIVE workgroups per CU, see plan_nucleotide -- whatever the tiles need)
        if constexpr (P == 6 && TW == 4 && !WIDE) rc = thrS ? launch_mt_thr<4, 2, MM, 5, 2>(d, st, a, nScout, nullptr, sp) : launch_mt_kernel<P, SW, SR, MM, 1, 2>(d, st, a, nScout);
        else rc = thrS ? launch_mt_kernel<P, TW, TRPL, MM, 4, 2>(d, st, a, nScout) : launch_mt_kernel<P, WIDE ? 16 : SW, WIDE ? 1 : SR, MM, 1, 2>(d, st, a, nScout);
        if (rc) return rc;
    }
    for (int r = 0; r < rounds; ++r) {
        FILL_TRY(flush_fills(d, st));
        hipLaunchKernelGGL(twl::mt_chain_kernel, dim3((n_run + 63) / 64), dim3(64), 0, st, (const int32_t *)d->mt_spath.p, sp_pitch, base.len, d_items, n_run,
                           (int32_t *)d->mt_chain.p, slots, marker, g_mt_perturb, (const int32_t *)a.mt_front);
        HIP_TRY(hipGetLastError());
        a.mt_jobs = (const int32_t *)d->mt_jobs.p + 3 * (size_t)nScout;
        if constexpr (kCanSmall) rc = thr ? (smallT ? launch_mt_thr<4, 2, MM, 5, 1>(d, st, a, nTile, grid_out, sp) : launch_mt_thr<TW, TRPL, MM, 4, 1>(d, st, a, nTile, grid_out, sp))
                                          : launch_mt_kernel<P, SW, SR, MM, 1, 1>(d, st, a, nTile, grid_out);
        else rc = thr ? launch_mt_kernel<P, TW, TRPL, MM, 4, 1>(d, st, a, nTile, grid_out) : launch_mt_kernel<P, SW, SR, MM, 1, 1>(d, st, a, nTile, grid_out);
        if (rc) return rc;
        a.mt_jobs = nullptr;
        a.mt_inline = (r == rounds - 1) ? 1 : 0;
        a.dbg = a.mt_inline ? base.dbg : nullptr;
        if ((rc = launch_mt_kernel<P, SW, SR, MM, 1, 3>(d, st, a, n_run))) return rc;
    }
    return TWL_OK;
}


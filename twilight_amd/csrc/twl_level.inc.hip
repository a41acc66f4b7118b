// twilight_amd/csrc/twl_level.inc.hip -- host side of the device-resident level API (include/twl_level.h).
// Included at the end of twl_align.hip: it shares that file's Device bookkeeping and run_device().
//
// HBM layout of a store (one device):
//   rows[2]    two planes of [n_seqs][cap] bytes; plane[i] says which one holds sequence i's current row.  A commit writes the new
//              row of every touched sequence into its other plane and flips the flag (SequenceInfo::changeStorage, sequencedb.cpp:51-55).
//   caches     one float[len][P] buffer per cached node profile (Node::msaFreq), addressed by the caller's ids.
//   level      raw[2n][stride][P] -> cols[2n][stride][P+2] (what the DP kernel reads), colinfo[2n][stride], lens, paths.

namespace {

struct CacheEntry {
    Buf buf;
    int32_t len = 0;
};

}  // namespace

// The small tables a call hands to its kernels (side descriptors, member lists, work lists ...) travel as ONE copy: they are laid out
// back to back in a pinned host block and land in a device block of the same layout.  One arena per kind of call (prepare / align /
// restore / commit): a call rewrites its host block only after an earlier synchronisation of the same store has seen the previous
// copy out of it complete, and the device block is rewritten in stream order behind the kernels that read the old content.
struct Ref { void *p = nullptr; };      // a table inside an arena
struct PinBuf {                         // pinned host memory a device-to-host copy lands in (no staging through pageable memory)
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return TWL_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        HIP_TRY(hipHostMalloc(&p, bytes + bytes / 2 + 256, hipHostMallocDefault));
        cap = bytes + bytes / 2 + 256;
        return TWL_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
struct Arena {
    char *h = nullptr;
    size_t hcap = 0, used = 0;
    Buf d;
    int begin(size_t bytes, size_t tables)      // room for `bytes` of payload in `tables` tables
    {
        const size_t want = bytes + 256 * (tables + 1);
        if (want > hcap) {
            if (h) (void)hipHostFree(h);
            h = nullptr; hcap = 0;
            const size_t cap = want + want / 2;
            HIP_TRY(hipHostMalloc((void **)&h, cap, hipHostMallocDefault));
            hcap = cap;
        }
        used = 0;
        return d.ensure(hcap);
    }
    template <class T>
    void put(Ref &r, const T *src, size_t n)
    {
        used = (used + 255) & ~(size_t)255;
        if (n) memcpy(h + used, src, n * sizeof(T));
        r.p = (char *)d.p + used;
        used += std::max<size_t>(n * sizeof(T), 16);
    }
    template <class T> void put(Ref &r, const std::vector<T> &v) { put(r, v.data(), v.size()); }
    int flush(hipStream_t st)
    {
        if (used) HIP_TRY(hipMemcpyAsync(d.p, h, used, hipMemcpyHostToDevice, st));
        return TWL_OK;
    }
    void release() { if (h) (void)hipHostFree(h); h = nullptr; hcap = 0; d.release(); }
};

// The device buffers of one prepared level (raw and packed columns are GBs at the leaf level).  They belong to the DEVICE, not to the
// store: a store takes a set at twl_level_prepare and gives it back at twl_level_commit, so the runs of a process that align one after
// the other work in the same, already mapped memory (device memory a process has not touched before costs tens of ms per GB:
// tools/micro/alloc_cost.hip), and a run holds only its rows while it waits.  Stores that are between prepare and commit at the same
// time (device replicas of a test) get a set each.
struct LevelBufs {
    Arena up_prepare, up_align, up_restore, up_commit;
    PinBuf back;                                             // lengths coming back from prepare / restore
    Ref d_sides, d_mseq, d_mw, d_mplane, d_tab, d_num;       // up_prepare (d_mplane, d_tab: up_commit after the commit's upload)
    Ref d_lenmask;                                           // up_align
    Ref r_sel;                                               // up_restore
    Ref d_pathlen, d_work, d_merge, d_mergew, d_fromdp;      // up_commit
    Buf d_raw, d_colinfo, d_cols, d_len, d_aln, d_alnlen, d_err;
    Buf d_paths, d_chunk, d_ccnt;
    Buf r_oidx, r_run, r_seg, r_aoff, r_blist, r_nboth, r_wtot, r_arena, r_outlen, r_tb, r_rows;      // twl_level_restore (restore_kernels.hip.h)
    Buf x_send, x_recv, x_rowoff, x_blkoff, x_len;                                  // exchange of final paths between processes (device blocks)
    bool busy = false;
    hipEvent_t ev_scan = nullptr, ev_apply = nullptr;        // a small level's row rewrite runs on the device's second stream: path_scan done / rewrite done
    bool apply_pending = false;                              // ... and may still read this set's tables
    void release_all()
    {
        if (ev_scan) (void)hipEventDestroy(ev_scan);
        if (ev_apply) (void)hipEventDestroy(ev_apply);
        ev_scan = ev_apply = nullptr; apply_pending = false;
        for (Buf *b : {&d_raw, &d_colinfo, &d_cols, &d_len, &d_aln, &d_alnlen, &d_err,
                       &d_paths, &d_chunk, &d_ccnt, &r_oidx, &r_run, &r_seg, &r_aoff, &r_blist, &r_nboth, &r_wtot, &r_arena, &r_outlen, &r_tb, &r_rows, &x_send, &x_recv, &x_rowoff, &x_blkoff, &x_len})
            b->release();
        for (Arena *a : {&up_prepare, &up_align, &up_restore, &up_commit}) a->release();
        back.release();
    }
};

namespace {
std::vector<LevelBufs *> &level_pool(Device *d)
{
    static std::mutex mu;
    static std::unordered_map<Device *, std::vector<LevelBufs *>> pools;
    std::lock_guard<std::mutex> lk(mu);
    return pools[d];
}
// (callers hold d->mu)
LevelBufs *acquire_level(Device *d)
{
    auto &pool = level_pool(d);
    LevelBufs *waiting = nullptr;        // free, but the row rewrite of its last level may still read its tables
    for (LevelBufs *b : pool) {
        if (b->busy) continue;
        if (b->apply_pending && hipEventQuery(b->ev_apply) != hipSuccess) { (void)hipGetLastError(); if (!waiting) waiting = b; continue; }
        b->apply_pending = false;
        b->busy = true;
        return b;
    }
    if (waiting && pool.size() >= 3) {   // (two sets alternate at the top of a tree; never more than three)
        (void)hipStreamWaitEvent(d->stream, waiting->ev_apply, 0);
        waiting->apply_pending = false;
        waiting->busy = true;
        return waiting;
    }
    auto *b = new LevelBufs();
    b->busy = true;
    pool.push_back(b);
    return b;
}
void release_level(LevelBufs *&lv) { if (lv) { lv->busy = false; lv = nullptr; } }

// Cached node profiles come and go with every level near the top of a tree (a merged profile replaces its two parts): their buffers are
// recycled through the device instead of hipMalloc / hipFree (which waits for the device) in the middle of a level.  Everything that
// touches them runs on the device's one stream, so a buffer may be handed out again while the kernel that last read it is still queued.
std::vector<Buf> &cache_pool(Device *d)
{
    static std::mutex mu;
    static std::unordered_map<Device *, std::vector<Buf>> pools;
    std::lock_guard<std::mutex> lk(mu);
    return pools[d];
}
// (callers hold d->mu)
int cache_buf_get(Device *d, Buf &b, size_t bytes)
{
    auto &pool = cache_pool(d);
    size_t best = pool.size();
    for (size_t k = 0; k < pool.size(); ++k)
        if (pool[k].cap >= bytes && (best == pool.size() || pool[k].cap < pool[best].cap)) best = k;
    if (best != pool.size()) { b = pool[best]; pool[best] = pool.back(); pool.pop_back(); return TWL_OK; }
    b = Buf{};
    return b.ensure(bytes + bytes / 4);      // (the next profile up the tree is a little longer)
}
void cache_buf_put(Device *d, Buf &b)
{
    if (!b.p) return;
    auto &pool = cache_pool(d);
    if (pool.size() >= 4096) {               // keep the larger ones (hipFree waits for the device -- for the row rewrite that is still running: a 64-entry pool
                                             // made every top-level commit of a 100 000-leaf tree wait 1-2.6 ms here)
        size_t small = 0;
        for (size_t k = 1; k < pool.size(); ++k) if (pool[k].cap < pool[small].cap) small = k;
        if (pool[small].cap < b.cap) std::swap(pool[small], b);
        b.release();
        return;
    }
    pool.push_back(b);
    b = Buf{};
}
}  // namespace

static void twl_level_pool_release(Device *d)
{
    auto &pool = level_pool(d);
    for (LevelBufs *b : pool) { b->release_all(); delete b; }
    pool.clear();
    for (Buf &b : cache_pool(d)) b.release();
    cache_pool(d).clear();
}

struct twl_store {
    Device *d = nullptr;
    int P = 6;
    char type = 'n';
    int32_t n_seqs = 0;
    int64_t cap = 0;
    Buf rows[2];
    std::vector<uint8_t> plane;
    std::vector<int32_t> len;
    std::unordered_map<int32_t, CacheEntry *> cache;
    Buf lut;
    // state of the level between prepare / align / commit
    int32_t n_pairs = 0, seq_len = 0;
    bool prepared = false;
    std::vector<twl_side> sides;
    std::vector<int32_t> members;
    std::vector<int32_t> h_len, h_num;
    std::vector<int32_t> h_work, h_merge;    // scratch of the commit (kept for its capacity)
    std::vector<float> h_mergew;
    std::vector<float *> h_tab;
    std::vector<uint8_t> h_mplane;
    bool commit_pending = false;             // the last commit's kernels may still run (its events are d->ev[6], d->ev[7])
    bool commit_side = false;                // ... its row rewrite on the second stream (timed by d->ev2)
    hipEvent_t rows_event = nullptr;         // != nullptr: a row rewrite on the second stream may still run; whoever reads or rewrites rows on the first stream waits for it
    LevelBufs *lv = nullptr;     // the level's device buffers, held from prepare to commit (from the device's pool, see LevelBufs)
    int32_t staged_stride = 0;   // > 0: twl_level_restore put this level's DP paths (and the restored ones) into lv->d_paths at this row pitch
    Buf d_gather, d_off, d_plane, d_rowlen;
    Buf x_send, x_recv;          // device blocks of the subtree exchange of a sharded run (twl_store_exchange_buffers)
    double prepare_ms = 0, commit_ms = 0;
};

namespace {

// letterIdx(type, toupper(c)) -- reference src/scoring-matrix.cpp:26-79
void build_lut(char type, uint8_t *lut)
{
    for (int c = 0; c < 256; ++c) {
        const int u = (c >= 'a' && c <= 'z') ? c - 32 : c;
        int v;
        if (type == 'n') {
            switch (u) {
            case 'A': v = 0; break;
            case 'C': v = 1; break;
            case 'G': v = 2; break;
            case 'T': case 'U': v = 3; break;
            case '-': case '.': v = 5; break;
            default: v = 4; break;
            }
        } else {
            static const char acids[] = "ACDEFGHIKLMNPQRSTVWY";
            v = 20;
            for (int k = 0; k < 20; ++k) if (u == acids[k]) v = k;
            if (u == '-' || u == '.') v = 21;
        }
        lut[c] = (uint8_t)v;
    }
}

// Room for the rows.  The final alignment is several times longer than the sequences (6.6x at 10 000 x 10 kbp, 22x on the synthetic
// 100 000 x 1.6 kbp family), and re-pitching costs more than its copy: device memory the process has not touched before comes at tens of
// ms per GB (tools/micro/alloc_cost.hip: a first 16 GB allocation 1.3 s, recycled ones < 1 ms), which a pass must not pay in its middle.
// So the planes start at 8x-48x the longest sequence (more for more sequences, see twl_store_create) and grow by half when they have to -- within a budget of a sixth of the device memory
// for both planes (288 GB of HBM are there to be used), never below what is needed.
int64_t rows_budget_cap(twl_store *s)
{
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return INT64_MAX; }
    const int64_t budget = (int64_t)(totalB / 6);
    return std::max<int64_t>(256, budget / (2 * std::max<int64_t>(1, s->n_seqs)));
}

int g_fail_next_row_allocs = 0;      // twl_set_knob(TWL_KNOB_FAIL_ROW_ALLOCS, n): the next n allocations of grow_rows fail (test of its fallback)

int grow_rows(twl_store *s, int64_t need, int64_t want = 0)
{
    if (need <= s->cap) return TWL_OK;
    Device *d = s->d;
    HIP_TRY(hipStreamSynchronize(d->stream2));      // (a row rewrite of the previous level may still run there)
    s->rows_event = nullptr;
    const int64_t minCap = (need + 255) & ~(int64_t)255;
    int64_t ncap = std::max(need, std::min(std::max(want, need + need / 2), rows_budget_cap(s)));
    ncap = (ncap + 255) & ~(int64_t)255;
    // Both planes must end up with ONE pitch: both new buffers are allocated before either plane is touched; if either allocation
    // fails at the generous pitch, both are given back and both are retried at the pitch that is needed.
    Buf nb[2];
    auto alloc_both = [&](int64_t cap) {
        for (int pl = 0; pl < 2; ++pl) {
            int rc = TWL_ERR_HIP;
            if (g_fail_next_row_allocs > 0) { --g_fail_next_row_allocs; g_err = "row allocation failed (test knob)"; }
            else rc = nb[pl].ensure((size_t)s->n_seqs * (size_t)cap);
            if (rc) { nb[0].release(); nb[1].release(); (void)hipGetLastError(); return rc; }      // (the failed hipMalloc's error is not left behind)
        }
        return (int)TWL_OK;
    };
    int rc = alloc_both(ncap);
    if (rc && ncap > minCap) { ncap = minCap; rc = alloc_both(ncap); }
    if (rc) return rc;
    for (int pl = 0; pl < 2; ++pl)
        if (s->rows[pl].p)
            HIP_TRY(hipMemcpy2DAsync(nb[pl].p, (size_t)ncap, s->rows[pl].p, (size_t)s->cap, (size_t)s->cap, (size_t)s->n_seqs, hipMemcpyDeviceToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (int pl = 0; pl < 2; ++pl) { s->rows[pl].release(); s->rows[pl] = nb[pl]; }
    s->cap = ncap;
    return TWL_OK;
}

// (callers hold s->d->mu)
void store_destroy_locked(twl_store *s)
{
    (void)hipSetDevice(s->d->id);
    (void)hipStreamSynchronize(s->d->stream);        // (a commit does not wait for its kernels)
    (void)hipStreamSynchronize(s->d->stream2);
    for (auto &kv : s->cache) { cache_buf_put(s->d, kv.second->buf); delete kv.second; }
    release_level(s->lv);
    for (Buf *b : {&s->rows[0], &s->rows[1], &s->lut, &s->d_gather, &s->d_off, &s->d_plane, &s->d_rowlen, &s->x_send, &s->x_recv})
        b->release();
    s->d->live_stores -= 1;
    delete s;
}

template <class T>
int upload(Buf &b, const std::vector<T> &v, hipStream_t st)
{
    int rc = b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return TWL_OK;
}

}  // namespace

extern "C" {

int twl_store_create(int device, char type, int32_t n_seqs, const char *const *seqs, const int32_t *lens, twl_store **out)
{
    if (!g_init) { g_err = "twl_init not called"; return TWL_ERR_NOT_INITIALIZED; }
    if (!out || n_seqs < 0 || (n_seqs > 0 && (!seqs || !lens)) || (type != 'n' && type != 'p')) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = nullptr;
    int rc = find_dev(device, &d);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    auto *s = new twl_store();
    d->live_stores += 1;
    s->d = d;
    s->type = type;
    s->P = (type == 'n') ? 6 : 22;
    s->n_seqs = n_seqs;
    s->plane.assign((size_t)n_seqs, 0);
    s->len.assign(lens, lens + n_seqs);
    int64_t maxLen = 1;
    for (int32_t i = 0; i < n_seqs; ++i) {
        if (lens[i] < 0) { delete s; g_err = "negative sequence length"; return TWL_ERR_BAD_ARGUMENT; }
        maxLen = std::max<int64_t>(maxLen, lens[i]);
    }
    // room for the alignment to grow (see grow_rows); the sequences go up through a tight host image with its own pitch
    // (how much longer than its sequences an alignment gets grows with the number of sequences: 6.6x at 10 000, 22x at 100 000 on the
    // synthetic families; 8 x log10(n) - 16, between 8x and 48x, within grow_rows' budget)
    const double lg = std::log10((double)std::max<int32_t>(n_seqs, 10));
    const int64_t factor = (int64_t)std::min(48.0, std::max(8.0, 8.0 * lg - 16.0));
    if ((rc = grow_rows(s, maxLen + 1, factor * maxLen + 256))) { store_destroy_locked(s); return rc; }
    if (n_seqs > 0) {
        const size_t hp = (size_t)maxLen;
        std::unique_ptr<char[]> img(new char[(size_t)n_seqs * hp]);
        for (int32_t i = 0; i < n_seqs; ++i) { memcpy(&img[(size_t)i * hp], seqs[i], (size_t)lens[i]); memset(&img[(size_t)i * hp + (size_t)lens[i]], '-', hp - (size_t)lens[i]); }
        HIP_TRY(hipMemcpy2D(s->rows[0].p, (size_t)s->cap, img.get(), hp, hp, (size_t)n_seqs, hipMemcpyHostToDevice));
    }
    uint8_t lut[256];
    build_lut(type, lut);
    if ((rc = s->lut.ensure(256))) { store_destroy_locked(s); return rc; }
    HIP_TRY(hipMemcpy(s->lut.p, lut, 256, hipMemcpyHostToDevice));
    *out = s;
    return TWL_OK;
}

void twl_store_destroy(twl_store *s)
{
    if (!s) return;
    std::lock_guard<std::mutex> lk(s->d->mu);      // (the level-buffer pool and the store count belong to the device)
    store_destroy_locked(s);
}

int twl_store_read_rows(twl_store *s, char *const *rows_out, int32_t *lens_out)
{
    if (!s || !lens_out) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    for (int32_t i = 0; i < s->n_seqs; ++i) lens_out[i] = s->len[i];
    if (!rows_out || s->n_seqs == 0) return TWL_OK;
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    std::vector<int64_t> off((size_t)s->n_seqs);
    int64_t total = 0;
    int32_t maxLen = 1;
    for (int32_t i = 0; i < s->n_seqs; ++i) { off[i] = total; total += s->len[i]; maxLen = std::max(maxLen, s->len[i]); }
    int rc;
    if ((rc = s->d_gather.ensure((size_t)std::max<int64_t>(total, 16)))) return rc;
    if (s->rows_event) { HIP_TRY(hipStreamWaitEvent(d->stream, s->rows_event, 0)); s->rows_event = nullptr; }
    if ((rc = upload(s->d_off, off, d->stream))) return rc;
    if ((rc = upload(s->d_plane, s->plane, d->stream))) return rc;
    if ((rc = upload(s->d_rowlen, s->len, d->stream))) return rc;
    hipLaunchKernelGGL(twl::gather_rows_kernel, dim3((unsigned)s->n_seqs, (unsigned)((maxLen + 255) / 256)), dim3(256), 0, d->stream,
                       (const char *)s->rows[0].p, (const char *)s->rows[1].p, s->cap, (const uint8_t *)s->d_plane.p, (const int32_t *)s->d_rowlen.p,
                       (const int64_t *)s->d_off.p, (char *)s->d_gather.p);
    HIP_TRY(hipGetLastError());
    // rows laid out back to back by the caller (row i at the prefix sum of the lengths): one transfer, straight into them
    bool packed = true;
    for (int32_t i = 0; i < s->n_seqs && packed; ++i) packed = (s->len[i] == 0) || (rows_out[i] == rows_out[0] + (off[i] - off[0]) && rows_out[0] != nullptr);
    if (packed && s->len[0] > 0) {
        if (total) HIP_TRY(hipMemcpyAsync(rows_out[0], s->d_gather.p, (size_t)total, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        return TWL_OK;
    }
    std::unique_ptr<char[]> host(new char[(size_t)std::max<int64_t>(total, 1)]);      // uninitialised on purpose
    if (total) HIP_TRY(hipMemcpyAsync(host.get(), s->d_gather.p, (size_t)total, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    // scatter into the caller's rows on a few threads (hundreds of MB at the end of a large run)
    const int nt = (int)std::min<size_t>(8, std::max<size_t>(1, (size_t)total >> 24));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (int32_t i = t; i < s->n_seqs; i += nt)
                if (rows_out[i] && s->len[i] > 0) memcpy(rows_out[i], &host[(size_t)off[i]], (size_t)s->len[i]);
        });
    for (auto &x : th) x.join();
    return TWL_OK;
}

// rows of a list of sequences: out = their current rows back to back (row t at the prefix sum of lens_out), lens_out[t] their lengths; out NULL: lengths only
int twl_store_read_rows_of(twl_store *s, int32_t n_ids, const int32_t *ids, char *out, int32_t *lens_out)
{
    if (!s || n_ids < 0 || (n_ids > 0 && (!ids || !lens_out))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    std::vector<int64_t> off((size_t)n_ids);
    std::vector<int32_t> ln((size_t)n_ids);
    int64_t total = 0;
    int32_t maxLen = 1;
    for (int32_t t = 0; t < n_ids; ++t) {
        if (ids[t] < 0 || ids[t] >= s->n_seqs) { g_err = "sequence id out of range"; return TWL_ERR_BAD_ARGUMENT; }
        ln[t] = lens_out[t] = s->len[ids[t]]; off[t] = total; total += ln[t]; maxLen = std::max(maxLen, ln[t]);
    }
    if (!out || n_ids == 0 || total == 0) return TWL_OK;
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    int rc;
    if ((rc = s->d_gather.ensure((size_t)total + (size_t)n_ids * 4 + 64))) return rc;
    if (s->rows_event) { HIP_TRY(hipStreamWaitEvent(d->stream, s->rows_event, 0)); s->rows_event = nullptr; }
    if ((rc = upload(s->d_off, off, d->stream))) return rc;
    if ((rc = upload(s->d_plane, s->plane, d->stream))) return rc;
    if ((rc = upload(s->d_rowlen, ln, d->stream))) return rc;
    int32_t *d_ids = reinterpret_cast<int32_t *>((char *)s->d_gather.p + (((size_t)total + 15) & ~(size_t)15));
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    hipLaunchKernelGGL(twl::gather_rows_of_kernel, dim3((unsigned)n_ids, (unsigned)((maxLen + 255) / 256)), dim3(256), 0, d->stream,
                       (const char *)s->rows[0].p, (const char *)s->rows[1].p, s->cap, (const uint8_t *)s->d_plane.p, (const int32_t *)d_ids, (const int32_t *)s->d_rowlen.p,
                       (const int64_t *)s->d_off.p, (char *)s->d_gather.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, s->d_gather.p, (size_t)total, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return TWL_OK;
}

// the inverse: rows (back to back in `in`, lengths lens) become the current rows of these sequences
int twl_store_write_rows(twl_store *s, int32_t n_ids, const int32_t *ids, const char *in, const int32_t *lens)
{
    if (!s || n_ids < 0 || (n_ids > 0 && (!ids || !lens || !in))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    if (n_ids == 0) return TWL_OK;
    std::vector<int64_t> off((size_t)n_ids);
    std::vector<int32_t> ln(lens, lens + n_ids);
    int64_t total = 0;
    int32_t maxLen = 1;
    for (int32_t t = 0; t < n_ids; ++t) {
        if (ids[t] < 0 || ids[t] >= s->n_seqs || lens[t] < 0) { g_err = "sequence id or length out of range"; return TWL_ERR_BAD_ARGUMENT; }
        off[t] = total; total += lens[t]; maxLen = std::max(maxLen, lens[t]);
    }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    int rc;
    if (s->rows_event) { HIP_TRY(hipStreamWaitEvent(d->stream, s->rows_event, 0)); s->rows_event = nullptr; }
    if ((rc = grow_rows(s, (int64_t)maxLen + 1))) return rc;
    if ((rc = s->d_gather.ensure((size_t)total + (size_t)n_ids * 4 + 64))) return rc;
    if ((rc = upload(s->d_off, off, d->stream))) return rc;
    if ((rc = upload(s->d_plane, s->plane, d->stream))) return rc;
    if ((rc = upload(s->d_rowlen, ln, d->stream))) return rc;
    int32_t *d_ids = reinterpret_cast<int32_t *>((char *)s->d_gather.p + (((size_t)total + 15) & ~(size_t)15));
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    if (total) HIP_TRY(hipMemcpyAsync(s->d_gather.p, in, (size_t)total, hipMemcpyHostToDevice, d->stream));
    hipLaunchKernelGGL(twl::scatter_rows_of_kernel, dim3((unsigned)n_ids, (unsigned)((maxLen + 255) / 256)), dim3(256), 0, d->stream,
                       (char *)s->rows[0].p, (char *)s->rows[1].p, s->cap, (const uint8_t *)s->d_plane.p, (const int32_t *)d_ids, (const int32_t *)s->d_rowlen.p,
                       (const int64_t *)s->d_off.p, (const char *)s->d_gather.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (int32_t t = 0; t < n_ids; ++t) s->len[ids[t]] = lens[t];
    return TWL_OK;
}

// The same two on DEVICE blocks (the rows of a sharded run's subtrees travel HBM to HBM): rows_to_block packs the current rows of `ids` back to back at
// dev_block (lens_out their lengths), rows_from_block makes the rows found there the current rows.  twl_store_exchange_buffers: two device buffers of the store.
static int rows_block(twl_store *s, bool toBlock, int32_t n_ids, const int32_t *ids, const int32_t *lens, void *dev_block)
{
    std::vector<int64_t> off((size_t)n_ids);
    std::vector<int32_t> ln(lens, lens + n_ids);
    int64_t total = 0;
    int32_t maxLen = 1;
    for (int32_t t = 0; t < n_ids; ++t) { off[t] = total; total += lens[t]; maxLen = std::max(maxLen, lens[t]); }
    Device *d = s->d;
    int rc;
    if (s->rows_event) { HIP_TRY(hipStreamWaitEvent(d->stream, s->rows_event, 0)); s->rows_event = nullptr; }
    if (!toBlock && (rc = grow_rows(s, (int64_t)maxLen + 1))) return rc;
    if ((rc = s->d_gather.ensure((size_t)n_ids * 4 + 64))) return rc;
    if ((rc = upload(s->d_off, off, d->stream))) return rc;
    if ((rc = upload(s->d_plane, s->plane, d->stream))) return rc;
    if ((rc = upload(s->d_rowlen, ln, d->stream))) return rc;
    HIP_TRY(hipMemcpyAsync(s->d_gather.p, ids, (size_t)n_ids * sizeof(int32_t), hipMemcpyHostToDevice, d->stream));
    const dim3 grid((unsigned)n_ids, (unsigned)((maxLen + 255) / 256));
    if (toBlock)
        hipLaunchKernelGGL(twl::gather_rows_of_kernel, grid, dim3(256), 0, d->stream, (const char *)s->rows[0].p, (const char *)s->rows[1].p, s->cap, (const uint8_t *)s->d_plane.p,
                           (const int32_t *)s->d_gather.p, (const int32_t *)s->d_rowlen.p, (const int64_t *)s->d_off.p, (char *)dev_block);
    else
        hipLaunchKernelGGL(twl::scatter_rows_of_kernel, grid, dim3(256), 0, d->stream, (char *)s->rows[0].p, (char *)s->rows[1].p, s->cap, (const uint8_t *)s->d_plane.p,
                           (const int32_t *)s->d_gather.p, (const int32_t *)s->d_rowlen.p, (const int64_t *)s->d_off.p, (const char *)dev_block);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream));
    if (!toBlock) for (int32_t t = 0; t < n_ids; ++t) s->len[ids[t]] = lens[t];
    return TWL_OK;
}
int twl_store_rows_to_block(twl_store *s, int32_t n_ids, const int32_t *ids, void *dev_block, int32_t *lens_out)
{
    if (!s || n_ids < 0 || (n_ids > 0 && (!ids || !lens_out || !dev_block))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    for (int32_t t = 0; t < n_ids; ++t) {
        if (ids[t] < 0 || ids[t] >= s->n_seqs) { g_err = "sequence id out of range"; return TWL_ERR_BAD_ARGUMENT; }
        lens_out[t] = s->len[ids[t]];
    }
    if (n_ids == 0) return TWL_OK;
    std::lock_guard<std::mutex> lk(s->d->mu);
    HIP_TRY(hipSetDevice(s->d->id));
    return rows_block(s, true, n_ids, ids, lens_out, dev_block);
}
int twl_store_rows_from_block(twl_store *s, int32_t n_ids, const int32_t *ids, const int32_t *lens, const void *dev_block)
{
    if (!s || n_ids < 0 || (n_ids > 0 && (!ids || !lens || !dev_block))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    for (int32_t t = 0; t < n_ids; ++t) if (ids[t] < 0 || ids[t] >= s->n_seqs || lens[t] < 0) { g_err = "sequence id or length out of range"; return TWL_ERR_BAD_ARGUMENT; }
    if (n_ids == 0) return TWL_OK;
    std::lock_guard<std::mutex> lk(s->d->mu);
    HIP_TRY(hipSetDevice(s->d->id));
    return rows_block(s, false, n_ids, ids, lens, const_cast<void *>(dev_block));
}
int twl_store_exchange_buffers(twl_store *s, int64_t send_bytes, int64_t recv_bytes, void **send_dev, void **recv_dev)
{
    if (!s || send_bytes < 0 || recv_bytes < 0 || !send_dev || !recv_dev) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    std::lock_guard<std::mutex> lk(s->d->mu);
    HIP_TRY(hipSetDevice(s->d->id));
    int rc;
    if ((rc = s->x_send.ensure((size_t)std::max<int64_t>(send_bytes, 16)))) return rc;
    if ((rc = s->x_recv.ensure((size_t)std::max<int64_t>(recv_bytes, 16)))) return rc;
    *send_dev = s->x_send.p; *recv_dev = s->x_recv.p;
    return TWL_OK;
}

// a cached profile arriving from another rank: float[len][P] under a NEW id of this store
int twl_store_write_cache(twl_store *s, int32_t id, const float *data, int32_t len)
{
    if (!s || !data || len < 0 || id < 0) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    if (s->cache.count(id)) { g_err = "cache id in use"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    auto *e = new CacheEntry();
    const size_t bytes = (size_t)len * s->P * sizeof(float);
    int rc = cache_buf_get(d, e->buf, std::max<size_t>(bytes, 16));
    if (rc) { delete e; return rc; }
    e->len = len;
    if (bytes) HIP_TRY(hipMemcpyAsync(e->buf.p, data, bytes, hipMemcpyHostToDevice, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    s->cache[id] = e;
    return TWL_OK;
}

int twl_store_read_cache(twl_store *s, int32_t id, float *out, int32_t *len_out)
{
    if (!s) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    auto it = s->cache.find(id);
    if (it == s->cache.end()) { g_err = "unknown cache id"; return TWL_ERR_BAD_ARGUMENT; }
    if (len_out) *len_out = it->second->len;
    if (!out) return TWL_OK;
    std::lock_guard<std::mutex> lk(s->d->mu);
    HIP_TRY(hipSetDevice(s->d->id));
    HIP_TRY(hipStreamSynchronize(s->d->stream));     // (a commit does not wait for its kernels)
    HIP_TRY(hipMemcpy(out, it->second->buf.p, (size_t)it->second->len * s->P * sizeof(float), hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_store_drop_cache(twl_store *s, int32_t id)
{
    if (!s) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    auto it = s->cache.find(id);
    if (it == s->cache.end()) return TWL_OK;
    std::lock_guard<std::mutex> lk(s->d->mu);
    (void)hipSetDevice(s->d->id);
    (void)hipStreamSynchronize(s->d->stream);
    cache_buf_put(s->d, it->second->buf);
    delete it->second;
    s->cache.erase(it);
    return TWL_OK;
}

int twl_level_prepare(twl_store *s, const twl_params *p, float gappy_threshold, int32_t n_pairs, const twl_side *sides, const int32_t *members,
                      const float *member_weight, int32_t seq_len, int32_t *len_out, uint8_t *colinfo_out)
{
    if (!s) { g_err = "store is null"; return TWL_ERR_BAD_ARGUMENT; }
    int rc = check_params(p);
    if (rc) return rc;
    if (p->P != s->P) { g_err = "params.P does not match the store's sequence type"; return TWL_ERR_BAD_ARGUMENT; }
    if (n_pairs < 0 || seq_len < 1 || (n_pairs > 0 && (!sides || !len_out))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    s->prepared = false;
    s->n_pairs = n_pairs;
    s->seq_len = seq_len;
    if (n_pairs == 0) { s->prepared = true; return TWL_OK; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    hipStream_t st = d->stream;
    const size_t ns = (size_t)n_pairs * 2, P = (size_t)s->P, sl = (size_t)seq_len;
    if (!s->lv) s->lv = acquire_level(d);      // held until the commit (or the store's end)

    s->sides.assign(sides, sides + ns);
    size_t nm = 0;
    for (size_t i = 0; i < ns; ++i) {
        const twl_side &sd = sides[i];
        if (sd.n_members < 0 || sd.member_off < 0 || sd.len < 0 || sd.len > seq_len || sd.num < 1) { g_err = "bad side descriptor"; return TWL_ERR_BAD_ARGUMENT; }
        nm = std::max(nm, (size_t)sd.member_off + (size_t)sd.n_members);
    }
    if (nm > 0 && (!members || !member_weight)) { g_err = "member tables missing"; return TWL_ERR_BAD_ARGUMENT; }
    s->members.assign(members, members + nm);
    std::vector<uint8_t> mplane(nm, 0);
    std::vector<twl::SideDesc> dsides(ns);
    std::vector<float *> tab;
    std::unordered_map<int32_t, int32_t> slotOf;
    auto slot = [&](int32_t id) {
        auto it = slotOf.find(id);
        if (it != slotOf.end()) return it->second;
        tab.push_back((float *)s->cache[id]->buf.p);
        return slotOf[id] = (int32_t)tab.size() - 1;
    };
    for (size_t i = 0; i < ns; ++i) {
        const twl_side &sd = sides[i];
        for (int32_t m = 0; m < sd.n_members; ++m) {
            const int32_t q = members[sd.member_off + m];
            if (q < 0 || q >= s->n_seqs) { g_err = "member sequence id out of range"; return TWL_ERR_BAD_ARGUMENT; }
            if (s->len[q] != sd.len) { g_err = "member row length differs from the side's len"; return TWL_ERR_BAD_ARGUMENT; }
            mplane[sd.member_off + m] = s->plane[q];
        }
        twl::SideDesc &ds = dsides[i];
        ds.n_members = sd.n_members; ds.member_off = sd.member_off; ds.len = sd.len; ds.num = sd.num; ds.weight = sd.weight;
        ds.cache_slot = ds.store_slot = -1; ds.pad = 0;
        if (sd.cache_id >= 0) {
            auto it = s->cache.find(sd.cache_id);
            if (it == s->cache.end() || it->second->len != sd.len) { g_err = "cache id unknown or of another length"; return TWL_ERR_BAD_ARGUMENT; }
            ds.cache_slot = slot(sd.cache_id);
        } else if (sd.store_id >= 0) {
            if (s->cache.count(sd.store_id)) { g_err = "store_id already in use"; return TWL_ERR_BAD_ARGUMENT; }
            auto *ce = new CacheEntry();
            ce->len = sd.len;
            if ((rc = cache_buf_get(d, ce->buf, std::max<size_t>((size_t)sd.len * P * sizeof(float), 16)))) { delete ce; return rc; }
            s->cache[sd.store_id] = ce;
            ds.store_slot = slot(sd.store_id);
        }
    }
    s->h_num.resize(ns);
    for (size_t i = 0; i < ns; ++i) s->h_num[i] = sides[i].num;

    if (s->rows_event) {       // a row rewrite on the second stream: sides that are built from rows (not from a cached profile) read what it writes
        bool readsRows = false;
        for (size_t i = 0; i < ns && !readsRows; ++i) readsRows = sides[i].cache_id < 0 && sides[i].n_members > 0;
        if (readsRows) { HIP_TRY(hipStreamWaitEvent(st, s->rows_event, 0)); s->rows_event = nullptr; }
    }
    HIP_TRY(hipEventRecord(d->ev[0], st));
    {
        Arena &A = s->lv->up_prepare;
        if ((rc = A.begin(ns * (sizeof(twl::SideDesc) + sizeof(int32_t)) + nm * (sizeof(int32_t) + sizeof(float) + 1) + tab.size() * sizeof(float *), 6))) return rc;
        A.put(s->lv->d_sides, dsides);
        A.put(s->lv->d_mseq, s->members);
        A.put(s->lv->d_mw, member_weight, nm);
        A.put(s->lv->d_mplane, mplane);
        A.put(s->lv->d_tab, tab);
        A.put(s->lv->d_num, s->h_num);
        if ((rc = A.flush(st))) return rc;
    }
    if ((rc = s->lv->d_raw.ensure(ns * sl * P * sizeof(float)))) return rc;
    if ((rc = s->lv->d_colinfo.ensure(ns * sl))) return rc;
    if ((rc = s->lv->d_cols.ensure(ns * sl * (P + 2) * sizeof(float)))) return rc;
    if ((rc = s->lv->d_len.ensure(ns * sizeof(int32_t)))) return rc;

    twl::LevelArgs a{};
    a.sides = (const twl::SideDesc *)s->lv->d_sides.p;
    a.member_seq = (const int32_t *)s->lv->d_mseq.p;
    a.member_w = (const float *)s->lv->d_mw.p;
    a.member_plane = (const uint8_t *)s->lv->d_mplane.p;
    a.rows0 = (const char *)s->rows[0].p; a.rows1 = (const char *)s->rows[1].p;
    a.cap = s->cap;
    a.cache = (float *const *)s->lv->d_tab.p;
    a.lut = (const uint8_t *)s->lut.p;
    a.raw = (float *)s->lv->d_raw.p;
    a.colinfo = (uint8_t *)s->lv->d_colinfo.p;
    a.cols = (float *)s->lv->d_cols.p;
    a.len_out = (int32_t *)s->lv->d_len.p;
    a.stride = seq_len;
    a.gappy_thr = gappy_threshold;
    a.remove = (gappy_threshold == 1.0) ? 0 : 1;                         // alignment-helper.cpp:77
    a.gap_open = p->gap_open; a.gap_extend = p->gap_extend;
    a.scale = (s->type == 'n') ? 0.5f : 1.0f;                            // :171
    a.min_gap_extend = p->gap_extend * 0.2;                              // :174  (double product narrowed to float)
    a.min_gap_open = p->gap_open * 0.1;                                  // :175
    int32_t maxLen = 1;
    for (size_t i = 0; i < ns; ++i) maxLen = std::max(maxLen, sides[i].len);
    const dim3 gridP((unsigned)ns, (unsigned)((maxLen + 1023) / 1024));
    if ((rc = s->lv->d_ccnt.ensure(ns * (size_t)gridP.y * sizeof(int32_t)))) return rc;
    a.chunk_cnt = (int32_t *)s->lv->d_ccnt.p;
    a.n_chunks = (int32_t)gridP.y;
    if (s->P == 6) {
        hipLaunchKernelGGL(twl::profile_kernel<6>, gridP, dim3(256), 0, st, a);
        hipLaunchKernelGGL(twl::compact_kernel<6>, gridP, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(twl::profile_kernel<22>, gridP, dim3(256), 0, st, a);
        hipLaunchKernelGGL(twl::compact_kernel<22>, gridP, dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d->ev[1], st));
    s->h_len.resize(ns);
    if ((rc = s->lv->back.ensure(ns * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(s->lv->back.p, s->lv->d_len.p, ns * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (colinfo_out) HIP_TRY(hipMemcpyAsync(colinfo_out, s->lv->d_colinfo.p, ns * sl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(s->h_len.data(), s->lv->back.p, ns * sizeof(int32_t));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[0], d->ev[1]));
    s->prepare_ms = ms;
    for (size_t i = 0; i < ns; ++i) len_out[i] = s->h_len[i];
    s->prepared = true;
    return TWL_OK;
}

int twl_level_read_colinfo(twl_store *s, int32_t pair, int32_t side, uint8_t *out)
{
    if (!s || !s->prepared || !out || pair >= s->n_pairs || side < 0 || side > 1) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    if (pair < 0) {      // the whole level: [n_pairs][2][seq_len]
        HIP_TRY(hipMemcpy(out, s->lv->d_colinfo.p, (size_t)s->n_pairs * 2 * (size_t)s->seq_len, hipMemcpyDeviceToHost));
        return TWL_OK;
    }
    const size_t idx = (size_t)pair * 2 + (size_t)side;
    const size_t len = (size_t)std::max(0, s->sides[idx].len);
    if (len) HIP_TRY(hipMemcpy(out, (const uint8_t *)s->lv->d_colinfo.p + idx * (size_t)s->seq_len, len, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_level_align(twl_store *s, const twl_params *p, const uint8_t *run_mask, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out)
{
    return twl_level_align_mixed(s, p, run_mask, nullptr, aln_out, aln_len_out, err_out);
}

// One call for the pairs of both gap-character kinds of a level (alignment-cpu.cpp:88 decides gapCharScore per pair): zero_gap[i] = 1 gives
// pair i gapCharScore 0, the others p->gap_char.
int twl_level_align_mixed(twl_store *s, const twl_params *p, const uint8_t *run_mask, const uint8_t *zero_gap, int8_t *aln_out, int32_t *aln_len_out, int16_t *err_out)
{
    if (!s || !s->prepared) { g_err = "twl_level_prepare has not been called"; return TWL_ERR_BAD_ARGUMENT; }
    int rc = check_params(p);
    if (rc) return rc;
    if (p->P != s->P) { g_err = "params.P does not match the store's sequence type"; return TWL_ERR_BAD_ARGUMENT; }
    const int32_t n = s->n_pairs;
    if (n == 0) return TWL_OK;
    if (!aln_len_out || !err_out) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    hipStream_t st = d->stream;
    const size_t sl = (size_t)s->seq_len;
    std::vector<int32_t> lm(s->h_len);
    if (run_mask)
        for (int32_t i = 0; i < n; ++i) if (!run_mask[i]) lm[2 * i] = lm[2 * i + 1] = 0;
    {
        Arena &A = s->lv->up_align;
        if ((rc = A.begin(lm.size() * sizeof(int32_t), 1))) return rc;
        A.put(s->lv->d_lenmask, lm);
        if ((rc = A.flush(st))) return rc;
    }
    if ((rc = s->lv->d_aln.ensure((size_t)n * 2 * sl))) return rc;
    if ((rc = s->lv->d_alnlen.ensure((size_t)n * sizeof(int32_t)))) return rc;
    if ((rc = s->lv->d_err.ensure((size_t)n * sizeof(int16_t)))) return rc;
    HIP_TRY(hipEventRecord(d->ev[5], st));
    // every selected query side a single, uncached sequence: its profile rows hold one letter each (profile_kernel built them)
    bool qryOneHot = true;
    for (int32_t i = 0; i < n && qryOneHot; ++i)
        if (lm[2 * i] > 0 && lm[2 * i + 1] > 0) qryOneHot = s->sides[2 * (size_t)i + 1].n_members == 1 && s->sides[2 * (size_t)i + 1].cache_id < 0;
    // what every selected pair looks like (run_device, shape): single uncached sequences on both sides -- no gap letter, denominators of 1
    int shape = 0;
    if (g_leaf_step) {
        bool leaf = true, any = false;
        for (int32_t i = 0; i < n && leaf; ++i) {
            if (!(lm[2 * i] > 0 && lm[2 * i + 1] > 0)) continue;
            any = true;
            for (int side = 0; side < 2; ++side) {
                const twl_side &sd = s->sides[2 * (size_t)i + side];
                leaf = leaf && sd.n_members == 1 && sd.cache_id < 0 && sd.num == 1;
            }
        }
        shape = (any && leaf) ? 2 : 0;
    }
    rc = run_device(d, st, p, n, s->seq_len, nullptr, nullptr, nullptr, (const int32_t *)s->lv->d_lenmask.p, (const int32_t *)s->lv->d_num.p, (int8_t *)s->lv->d_aln.p,
                    (int32_t *)s->lv->d_alnlen.p, (int16_t *)s->lv->d_err.p, lm.data(), (const float *)s->lv->d_cols.p, qryOneHot, zero_gap, shape);
    if (rc) return rc;
    if ((int32_t)d->last_err.size() == n && (int32_t)d->last_alnlen.size() == n) {      // (run_device read them back already, in its one synchronisation)
        std::copy(d->last_err.begin(), d->last_err.end(), err_out);
        std::copy(d->last_alnlen.begin(), d->last_alnlen.end(), aln_len_out);
        if (!aln_out) { d->stats.total_ms = d->stats.kernel_ms; return TWL_OK; }       // the paths stay in HBM (twl_level_read_path / twl_level_commit_from_dp)
    } else {
        HIP_TRY(hipMemcpyAsync(aln_len_out, s->lv->d_alnlen.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(err_out, s->lv->d_err.p, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    // paths: bulk when most pairs ran, else one copy per pair that has a path
    int32_t ran = 0;
    for (int32_t i = 0; i < n; ++i) ran += aln_len_out[i] > 0;
    if (!aln_out) {}
    else if (ran * 2 >= n) HIP_TRY(hipMemcpyAsync(aln_out, s->lv->d_aln.p, (size_t)n * 2 * sl, hipMemcpyDeviceToHost, st));
    else
        for (int32_t i = 0; i < n; ++i)
            if (aln_len_out[i] > 0)
                HIP_TRY(hipMemcpyAsync(aln_out + (size_t)i * 2 * sl, (int8_t *)s->lv->d_aln.p + (size_t)i * 2 * sl, (size_t)aln_len_out[i], hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(d->ev[3], st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[5], d->ev[3]));
    d->stats.total_ms = ms;
    return TWL_OK;
}

int twl_level_read_path(twl_store *s, int32_t pair, int8_t *out, int32_t len)
{
    if (!s || !s->prepared || !out || pair < 0 || pair >= s->n_pairs || len < 0 || len > 2 * s->seq_len || !s->lv || !s->lv->d_aln.p) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    if (len) HIP_TRY(hipMemcpy(out, (const int8_t *)s->lv->d_aln.p + (size_t)pair * 2 * (size_t)s->seq_len, (size_t)len, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_level_read_colinfo_many(twl_store *s, int32_t n_sel, const int32_t *pairs, uint8_t *out)
{
    if (!s || !s->prepared || n_sel < 0 || (n_sel > 0 && (!pairs || !out || !s->lv))) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    const size_t sl = (size_t)s->seq_len;
    for (int32_t t = 0; t < n_sel; ++t) {
        if (pairs[t] < 0 || pairs[t] >= s->n_pairs) { g_err = "pair index out of range"; return TWL_ERR_BAD_ARGUMENT; }
        HIP_TRY(hipMemcpyAsync(out + (size_t)t * 2 * sl, (const uint8_t *)s->lv->d_colinfo.p + (size_t)pairs[t] * 2 * sl, 2 * sl, hipMemcpyDeviceToHost, d->stream));
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
    return TWL_OK;
}

int twl_level_read_paths(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, int8_t *out, int32_t out_stride)
{
    if (!s || !s->prepared || n_sel < 0 || (n_sel > 0 && (!pairs || !lens || !out || !s->lv || !s->lv->d_aln.p)) || out_stride < 0) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    const size_t sl = (size_t)s->seq_len;
    for (int32_t t = 0; t < n_sel; ++t) {
        if (pairs[t] < 0 || pairs[t] >= s->n_pairs || lens[t] < 0 || lens[t] > out_stride || (size_t)lens[t] > 2 * sl) { g_err = "bad selection"; return TWL_ERR_BAD_ARGUMENT; }
        if (lens[t]) HIP_TRY(hipMemcpyAsync(out + (size_t)t * (size_t)out_stride, (const int8_t *)s->lv->d_aln.p + (size_t)pairs[t] * 2 * sl, (size_t)lens[t], hipMemcpyDeviceToHost, d->stream));
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
    return TWL_OK;
}

// One batch of twl_level_restore (the caller holds the device's lock and has sized the path buffer).
static int restore_batch(twl_store *s, const twl_params *p, int32_t n_sel, const int32_t *pairs, int32_t out_stride, int32_t *final_len_out)
{
    int rc;
    const int32_t n = s->n_pairs;
    Device *d = s->d;
    hipStream_t st = d->stream;
    LevelBufs *lv = s->lv;
    const size_t sl = (size_t)s->seq_len, bstride = 2 * sl + 1;
    const size_t ns = (size_t)n_sel;
    auto handBack = [&](int rcAlloc) {          // out of device memory for this batch: its pairs go back to the host path
        (void)hipGetLastError();
        for (int32_t t = 0; t < n_sel; ++t) final_len_out[t] = -1;
        (void)rcAlloc;
        return (int)TWL_OK;
    };
    {
        Arena &A = lv->up_restore;
        if ((rc = A.begin(ns * sizeof(int32_t), 1))) return rc;
        A.put(lv->r_sel, pairs, ns);
        if ((rc = A.flush(st))) return rc;
    }
    if ((rc = lv->r_oidx.ensure(2 * ns * (sl + 1) * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_run.ensure(ns * 4 * bstride * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_seg.ensure(ns * bstride * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_aoff.ensure(ns * bstride * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_blist.ensure(ns * bstride * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_nboth.ensure(ns * sizeof(int32_t)))) return rc;
    const unsigned nWch = (unsigned)((bstride + (size_t)twl::kRsThreads * twl::kWrItems - 1) / ((size_t)twl::kRsThreads * twl::kWrItems));      // chunks of boundaries of the longest possible path
    if ((rc = lv->r_wtot.ensure(ns * nWch * sizeof(int32_t)))) return handBack(rc);
    if ((rc = lv->r_arena.ensure(ns * (size_t)out_stride))) return handBack(rc);
    if ((rc = lv->r_outlen.ensure((size_t)n * sizeof(int32_t)))) return rc;
    const unsigned nb = (unsigned)std::max(1, std::min(128, 4096 / n_sel));     // one-wave workgroups per pair of the small alignments (<= ~1.8 GB of scratch for the rare large ones)
    if ((rc = lv->r_tb.ensure(ns * nb * twl::kNwThreads * (size_t)twl::kNwCells))) return handBack(rc);
    if ((rc = lv->r_rows.ensure(ns * nb * twl::kNwThreads * 6 * (size_t)twl::kNwRow * sizeof(float)))) return handBack(rc);
    twl::RestoreArgs a{};
    a.aln = (const int8_t *)lv->d_aln.p; a.aln_len = (const int32_t *)lv->d_alnlen.p; a.aln_stride = (int32_t)(2 * sl);
    a.colinfo = (const uint8_t *)lv->d_colinfo.p; a.stride = s->seq_len;
    a.sides = (const twl::SideDesc *)lv->d_sides.p; a.len_red = (const int32_t *)lv->d_len.p;
    a.sel = (const int32_t *)lv->r_sel.p; a.n_sel = n_sel;
    a.orig_idx = (int32_t *)lv->r_oidx.p; a.run = (int32_t *)lv->r_run.p; a.seg = (int32_t *)lv->r_seg.p; a.aoff = (int32_t *)lv->r_aoff.p;
    a.both_list = (int32_t *)lv->r_blist.p; a.n_both = (int32_t *)lv->r_nboth.p;
    a.wtot = (int32_t *)lv->r_wtot.p; a.n_wchunks = (int32_t)nWch;
    a.arena = (int8_t *)lv->r_arena.p; a.bstride = (int32_t)bstride;
    a.out = (int8_t *)lv->d_paths.p; a.out_stride = out_stride; a.out_len = (int32_t *)lv->r_outlen.p;
    a.tbs = (int8_t *)lv->r_tb.p; a.rows = (float *)lv->r_rows.p;
    a.ms = p->P - 1;
    for (int t = 0; t < a.ms * a.ms; ++t) a.M[t] = p->matrix[t];
    a.gap_open = p->gap_open; a.gap_extend = p->gap_extend;
    hipLaunchKernelGGL(twl::restore_index_kernel, dim3(2 * (unsigned)n_sel), dim3(twl::kRsThreads), 0, st, a);
    hipLaunchKernelGGL(twl::restore_runs_kernel, dim3((unsigned)n_sel), dim3(twl::kRsThreads), 0, st, a);
    hipLaunchKernelGGL(twl::restore_align_kernel, dim3((unsigned)n_sel, nb), dim3(twl::kNwThreads), 0, st, a);
    hipLaunchKernelGGL(twl::restore_count_kernel, dim3((unsigned)n_sel, nWch), dim3(twl::kRsThreads), 0, st, a);
    hipLaunchKernelGGL(twl::restore_write_kernel, dim3((unsigned)n_sel, nWch), dim3(twl::kRsThreads), 0, st, a);
    HIP_TRY(hipGetLastError());
    if ((rc = lv->back.ensure((size_t)n * sizeof(int32_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(lv->back.p, lv->r_outlen.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int32_t *all = (const int32_t *)lv->back.p;
    for (int32_t t = 0; t < n_sel; ++t) final_len_out[t] = all[pairs[t]];
    return TWL_OK;
}

// addGappyColumnsBack for the pairs `pairs` of the prepared and aligned level, on the device (restore_kernels.hip.h): their final paths go
// into the level's path buffer (row pitch out_stride, the pitch the commit must then be given), final_len_out[t] = the final length of
// pairs[t], or -1 when that pair holds a two-sided run too large for the device (the caller restores it on the host, as before).
int twl_level_restore(twl_store *s, const twl_params *p, int32_t n_sel, const int32_t *pairs, int32_t out_stride, int32_t *final_len_out)
{
    if (!s || !s->prepared || !s->lv || (n_sel > 0 && !s->lv->d_aln.p)) { g_err = "twl_level_align has not been called"; return TWL_ERR_BAD_ARGUMENT; }
    int rc = check_params(p);
    if (rc) return rc;
    const int32_t n = s->n_pairs;
    if (n_sel < 0 || (n_sel > 0 && (!pairs || !final_len_out)) || out_stride < 1) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    for (int32_t t = 0; t < n_sel; ++t) if (pairs[t] < 0 || pairs[t] >= n) { g_err = "pair index out of range"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    // the restored paths go straight into the rows of the level's path buffer (the commit leaves those rows alone); several calls per
    // level share the buffer and must agree on its pitch
    if (s->staged_stride && s->staged_stride != out_stride) { g_err = "twl_level_restore: one row pitch per level"; return TWL_ERR_BAD_ARGUMENT; }
    if ((rc = s->lv->d_paths.ensure((size_t)n * (size_t)out_stride))) return rc;
    s->staged_stride = out_stride;
    //  at most 4096 pairs per batch: the work arrays cost ~32 * (2 * seq_len + 1) bytes per pair plus the scratch of the small alignments
    // (~458 KB per pair once 4096 / n_sel floors at one workgroup); a wide level of a 100 000-leaf tree with a low -r would otherwise ask for tens of GB at
    // once.  A batch whose arrays cannot be had is handed back (-1: the caller restores those pairs on the host, a path that still exists).
    constexpr int32_t kBatch = 4096;
    for (int32_t at = 0; at < n_sel; at += kBatch)
        if ((rc = restore_batch(s, p, std::min(kBatch, n_sel - at), pairs + at, out_stride, final_len_out + at))) return rc;
    return TWL_OK;
}

// ---- the staged path buffer and contiguous device blocks (exchange of final paths between processes, include/twl_level.h) ----
int twl_level_exchange_buffers(twl_store *s, int64_t send_bytes, int64_t recv_bytes, void **send_dev, void **recv_dev)
{
    if (!s || !s->prepared || !s->lv || send_bytes < 0 || recv_bytes < 0 || !send_dev || !recv_dev) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    int rc;
    if ((rc = s->lv->x_send.ensure(std::max<size_t>((size_t)send_bytes, 64)))) return rc;
    if ((rc = s->lv->x_recv.ensure(std::max<size_t>((size_t)recv_bytes, 64)))) return rc;
    *send_dev = s->lv->x_send.p; *recv_dev = s->lv->x_recv.p;
    return TWL_OK;
}

static int level_rows_block(twl_store *s, bool toBlock, int32_t n_sel, const int32_t *pairs, const int32_t *lens, const uint8_t *where, void *blk_dev, const int64_t *blk_off)
{
    if (!s || !s->prepared || !s->lv || !s->staged_stride || n_sel < 0 || (n_sel > 0 && (!pairs || !lens || !blk_dev || !blk_off))) { g_err = "bad argument (twl_level_restore first)"; return TWL_ERR_BAD_ARGUMENT; }
    if (n_sel == 0) return TWL_OK;
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    hipStream_t st = d->stream;
    LevelBufs *lv = s->lv;
    const size_t sl2 = 2 * (size_t)s->seq_len, ps = (size_t)s->staged_stride;
    // rows of the DP output (where[t] == 1; only as a source) and rows of the staged path buffer (2), one launch per buffer
    for (int which = 1; which <= 2; ++which) {
        std::vector<int64_t> ro, bo;
        std::vector<int32_t> ln;
        for (int32_t t = 0; t < n_sel; ++t) {
            const int w = where ? where[t] : 2;
            if (w != which || lens[t] <= 0) continue;
            if (pairs[t] < 0 || pairs[t] >= s->n_pairs || (size_t)lens[t] > (which == 1 ? sl2 : ps) || (which == 1 && !toBlock)) { g_err = "bad row selection"; return TWL_ERR_BAD_ARGUMENT; }
            ro.push_back((int64_t)pairs[t] * (int64_t)(which == 1 ? sl2 : ps)); bo.push_back(blk_off[t]); ln.push_back(lens[t]);
        }
        if (ro.empty()) continue;
        int rc;
        if ((rc = upload(lv->x_rowoff, ro, st))) return rc;
        if ((rc = upload(lv->x_blkoff, bo, st))) return rc;
        if ((rc = upload(lv->x_len, ln, st))) return rc;
        int8_t *base = (int8_t *)(which == 1 ? lv->d_aln.p : lv->d_paths.p);
        if (toBlock) hipLaunchKernelGGL(twl::rows_to_block_kernel, dim3((unsigned)ro.size()), dim3(256), 0, st, (const int8_t *)base, (const int64_t *)lv->x_rowoff.p, (const int32_t *)lv->x_len.p, (int8_t *)blk_dev, (const int64_t *)lv->x_blkoff.p);
        else hipLaunchKernelGGL(twl::block_to_rows_kernel, dim3((unsigned)ro.size()), dim3(256), 0, st, base, (const int64_t *)lv->x_rowoff.p, (const int32_t *)lv->x_len.p, (const int8_t *)blk_dev, (const int64_t *)lv->x_blkoff.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));      // (the host vectors above go out of scope; the caller hands the block to a collective next)
    }
    return TWL_OK;
}

int twl_level_paths_to_block(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, const uint8_t *where, void *blk_dev, const int64_t *blk_off)
{
    return level_rows_block(s, true, n_sel, pairs, lens, where, blk_dev, blk_off);
}

int twl_level_paths_from_block(twl_store *s, int32_t n_sel, const int32_t *pairs, const int32_t *lens, const void *blk_dev, const int64_t *blk_off)
{
    return level_rows_block(s, false, n_sel, pairs, lens, nullptr, const_cast<void *>(blk_dev), blk_off);
}

int twl_level_write_final(twl_store *s, int32_t pair, const int8_t *path, int32_t len)
{
    if (!s || !s->prepared || !s->lv || !s->staged_stride || pair < 0 || pair >= s->n_pairs || len < 0 || len > s->staged_stride || (len > 0 && !path)) { g_err = "bad argument (twl_level_restore first)"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    if (len) HIP_TRY(hipMemcpy((int8_t *)s->lv->d_paths.p + (size_t)pair * (size_t)s->staged_stride, path, (size_t)len, hipMemcpyHostToDevice));
    return TWL_OK;
}

int twl_level_read_final(twl_store *s, int32_t pair, int8_t *out, int32_t len)
{
    if (!s || !s->prepared || !out || pair < 0 || pair >= s->n_pairs || len < 0 || !s->lv || !s->staged_stride || len > s->staged_stride) { g_err = "bad argument (twl_level_restore first)"; return TWL_ERR_BAD_ARGUMENT; }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    HIP_TRY(hipSetDevice(d->id));
    if (len) HIP_TRY(hipMemcpy(out, (const int8_t *)s->lv->d_paths.p + (size_t)pair * (size_t)s->staged_stride, (size_t)len, hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_level_commit(twl_store *s, const int8_t *paths, const int32_t *path_len, int32_t path_stride)
{
    return twl_level_commit_from_dp(s, paths, path_len, path_stride, nullptr);
}

int twl_level_commit_from_dp(twl_store *s, const int8_t *paths, const int32_t *path_len, int32_t path_stride, const uint8_t *from_dp)
{
    if (!s || !s->prepared) { g_err = "twl_level_prepare has not been called"; return TWL_ERR_BAD_ARGUMENT; }
    const int32_t n = s->n_pairs;
    const int32_t staged = s->staged_stride;
    // (the arguments are checked BEFORE the level's state is given up -- a rejected call leaves the level as it was, committable)
    if (n > 0 && ((!paths && !from_dp) || !path_len || path_stride < 1)) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    if (staged && (!from_dp || staged != path_stride)) { g_err = "commit after twl_level_restore: from_dp and the restore's row pitch are required"; return TWL_ERR_BAD_ARGUMENT; }
    if (from_dp && !staged) for (int32_t i = 0; i < n; ++i) if (from_dp[i] == 2) { g_err = "from_dp == 2 without twl_level_restore"; return TWL_ERR_BAD_ARGUMENT; }
    if (from_dp) {
        for (int32_t i = 0; i < n; ++i) {
            if (from_dp[i] == 1 && (!s->lv || !s->lv->d_aln.p || path_len[i] > 2 * s->seq_len)) { g_err = "from_dp without a DP output of this level"; return TWL_ERR_BAD_ARGUMENT; }
            if (!from_dp[i] && path_len[i] > 0 && !paths) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
        }
    }
    int32_t maxPath = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (path_len[i] < 0 || path_len[i] > path_stride) { g_err = "path_len outside [0, path_stride]"; return TWL_ERR_BAD_ARGUMENT; }
        maxPath = std::max(maxPath, path_len[i]);
    }
    Device *d = s->d;
    std::lock_guard<std::mutex> lk(d->mu);
    s->prepared = false;                       // the commit goes ahead: the level is over whatever happens from here on
    s->staged_stride = 0;
    struct GiveBack { twl_store *s; ~GiveBack() { release_level(s->lv); } } giveBack{s};      // the level's buffers return to the device's pool
    if (n == 0 || maxPath == 0) return TWL_OK;
    HIP_TRY(hipSetDevice(d->id));
    hipStream_t st = d->stream;
    const size_t P = (size_t)s->P;
    int rc;
    if ((rc = grow_rows(s, maxPath))) return rc;
    const int32_t nChunks = (maxPath + 255) / 256;

    constexpr int MG = 64;                      // members per workgroup of the row rewrite (its scan of the path chunk is shared by them)
    std::vector<int32_t> &work = s->h_work, &merge = s->h_merge;
    std::vector<float> &mergew = s->h_mergew;
    std::vector<float *> &tab = s->h_tab;
    work.clear(); merge.clear(); mergew.clear(); tab.clear();
    struct Pending { int32_t refId, qryId; CacheEntry *dst; };
    std::vector<Pending> pend;
    for (int32_t i = 0; i < n; ++i) {
        if (path_len[i] == 0) continue;
        for (int sd = 0; sd < 2; ++sd) {
            const twl_side &x = s->sides[2 * (size_t)i + sd];
            for (int32_t m = 0; m < x.n_members; m += MG) { work.push_back(2 * i + sd); work.push_back(m); work.push_back(std::min(MG, x.n_members - m)); }
        }
        const twl_side &r = s->sides[2 * (size_t)i], &q = s->sides[2 * (size_t)i + 1];
        const int32_t rid = r.cache_id >= 0 ? r.cache_id : r.store_id, qid = q.cache_id >= 0 ? q.cache_id : q.store_id;
        if (rid >= 0 && qid >= 0) {             // updateFrequency: both nodes carry a cached profile
            auto *ce = new CacheEntry();
            ce->len = path_len[i];
            if ((rc = cache_buf_get(d, ce->buf, (size_t)path_len[i] * P * sizeof(float)))) { delete ce; return rc; }
            merge.push_back(i);
            merge.push_back((int32_t)tab.size()); tab.push_back((float *)s->cache[rid]->buf.p);
            merge.push_back((int32_t)tab.size()); tab.push_back((float *)s->cache[qid]->buf.p);
            merge.push_back((int32_t)tab.size()); tab.push_back((float *)ce->buf.p);
            mergew.push_back(r.weight); mergew.push_back(q.weight);
            pend.push_back({rid, qid, ce});
        }
    }
    // current planes of the members (prepare's table may be stale if a sequence took part in an earlier commit of this level: it cannot,
    // a sequence belongs to one node of one pair per level)
    std::vector<uint8_t> &mplane = s->h_mplane;
    mplane.resize(s->members.size());
    for (size_t k = 0; k < s->members.size(); ++k) mplane[k] = s->plane[s->members[k]];

    bool hostRows = !from_dp;                  // rows of the caller's `paths` are uploaded: wait for them below
    HIP_TRY(hipEventRecord(d->ev[6], st));
    if ((rc = s->lv->d_paths.ensure((size_t)n * (size_t)path_stride))) return rc;
    if (!from_dp) HIP_TRY(hipMemcpyAsync(s->lv->d_paths.p, paths, (size_t)n * (size_t)path_stride, hipMemcpyHostToDevice, st));
    else {
        // the DP output of this level stays where the DP kernel left it (the commit kernels read row i of that buffer for from_dp[i] == 1; until
        // round 4 every run of such rows was copied into the path buffer: ~1700 small 2-D copies per pass of a 100 000-leaf tree); only the (few)
        // rows the caller brought are uploaded
        // (from_dp[i] == 2: twl_level_restore has put pair i's final path into its row already)
        for (int32_t i = 0; i < n; ++i)
            if (!from_dp[i] && path_len[i] > 0 && (hostRows = true))
                HIP_TRY(hipMemcpyAsync((int8_t *)s->lv->d_paths.p + (size_t)i * (size_t)path_stride, paths + (size_t)i * (size_t)path_stride, (size_t)path_len[i], hipMemcpyHostToDevice, st));
    }
    if ((rc = s->lv->d_chunk.ensure((size_t)n * nChunks * 2 * sizeof(int32_t)))) return rc;
    {
        Arena &A = s->lv->up_commit;
        if ((rc = A.begin(((size_t)n + work.size() + merge.size()) * sizeof(int32_t) + mergew.size() * sizeof(float) + tab.size() * sizeof(float *) + mplane.size() + (from_dp ? (size_t)n : 0), 7))) return rc;
        A.put(s->lv->d_pathlen, path_len, (size_t)n);
        A.put(s->lv->d_work, work);
        A.put(s->lv->d_merge, merge);
        A.put(s->lv->d_mergew, mergew);
        A.put(s->lv->d_tab, tab);
        A.put(s->lv->d_mplane, mplane);
        if (from_dp) A.put(s->lv->d_fromdp, from_dp, (size_t)n);
        if ((rc = A.flush(st))) return rc;
    }

    twl::CommitArgs a{};
    a.paths = (const int8_t *)s->lv->d_paths.p;
    if (from_dp) { a.from_dp = (const uint8_t *)s->lv->d_fromdp.p; a.paths_dp = (const int8_t *)s->lv->d_aln.p; a.dp_stride = (int32_t)(2 * (size_t)s->seq_len); }
    a.path_len = (const int32_t *)s->lv->d_pathlen.p;
    a.path_stride = path_stride;
    a.chunk_base = (int32_t *)s->lv->d_chunk.p;
    a.n_chunks = nChunks;
    a.sides = (const twl::SideDesc *)s->lv->d_sides.p;
    a.member_seq = (const int32_t *)s->lv->d_mseq.p;
    a.member_plane = (const uint8_t *)s->lv->d_mplane.p;
    a.rows0 = (char *)s->rows[0].p; a.rows1 = (char *)s->rows[1].p;
    a.cap = s->cap;
    a.work = (const int32_t *)s->lv->d_work.p;
    a.cache = (float *const *)s->lv->d_tab.p;
    a.merge = (const int32_t *)s->lv->d_merge.p;
    a.merge_w = (const float *)s->lv->d_mergew.p;
    hipLaunchKernelGGL(twl::path_scan_kernel, dim3((unsigned)n), dim3(256), 0, st, a);
    const unsigned nWork = (unsigned)(work.size() / 3), nMerge = (unsigned)(merge.size() / 4);
    // The row rewrite of a SMALL level (the top of a tree: a pair or a few, each with thousands of rows, 0.1-3 ms of HBM traffic) goes to
    // the device's second stream: the next levels there align a few pairs on a few CUs from cached profiles and need no row, so the rewrite
    // runs beside them.  Whatever reads or rewrites rows on the first stream waits for the event; the tables it reads belong to this level
    // set, which acquire_level does not hand out again before the rewrite is done (a second set takes the next level).
    const bool side = nWork > 0 && n <= 32 && !hostRows;
    s->commit_side = false;
    if (nWork && !side) {
        if (s->rows_event) { HIP_TRY(hipStreamWaitEvent(st, s->rows_event, 0)); s->rows_event = nullptr; }
        hipLaunchKernelGGL(twl::apply_path_kernel, dim3(nWork, (unsigned)((nChunks + 3) / 4)), dim3(256), 0, st, a);
    } else if (nWork) {
        LevelBufs *lv = s->lv;
        if (!lv->ev_scan) { HIP_TRY(hipEventCreateWithFlags(&lv->ev_scan, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&lv->ev_apply, hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(lv->ev_scan, st));
        HIP_TRY(hipStreamWaitEvent(d->stream2, lv->ev_scan, 0));        // (rewrites on the second stream follow each other in order)
        HIP_TRY(hipEventRecord(d->ev2[0], d->stream2));
        hipLaunchKernelGGL(twl::apply_path_kernel, dim3(nWork, (unsigned)((nChunks + 3) / 4)), dim3(256), 0, d->stream2, a);
        HIP_TRY(hipEventRecord(d->ev2[1], d->stream2));
        HIP_TRY(hipEventRecord(lv->ev_apply, d->stream2));
        lv->apply_pending = true;
        s->rows_event = lv->ev_apply;
        s->commit_side = true;
    }
    if (nMerge) {
        if (s->P == 6) hipLaunchKernelGGL(twl::merge_cache_kernel<6>, dim3(nMerge, (unsigned)nChunks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(twl::merge_cache_kernel<22>, dim3(nMerge, (unsigned)nChunks), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(d->ev[7], st));
    s->commit_pending = true;
    // No wait for the kernels: whatever touches the rows, the caches or the level's buffers next is queued behind them on the device's
    // stream, and the host's share of the next level overlaps with the row rewrite.  (Only the caller's own `paths` must be consumed.)
    if (hostRows) HIP_TRY(hipStreamSynchronize(st));

    // bookkeeping: every member of a committed pair now lives in its other plane with the path's length
    for (int32_t i = 0; i < n; ++i) {
        if (path_len[i] == 0) continue;
        for (int sd = 0; sd < 2; ++sd) {
            const twl_side &x = s->sides[2 * (size_t)i + sd];
            for (int32_t m = 0; m < x.n_members; ++m) {
                const int32_t q = s->members[x.member_off + m];
                s->plane[q] ^= 1;
                s->len[q] = path_len[i];
            }
        }
    }
    for (const Pending &pe : pend) {            // merged profile replaces the reference node's cache, the query node's is dropped
        CacheEntry *oldR = s->cache[pe.refId], *oldQ = s->cache[pe.qryId];
        cache_buf_put(d, oldR->buf); delete oldR;      // (merge_cache_kernel may still read them: the pool hands them out on the same stream)
        cache_buf_put(d, oldQ->buf); delete oldQ;
        s->cache.erase(pe.qryId);
        s->cache[pe.refId] = pe.dst;
    }
    return TWL_OK;
}

int twl_level_read_columns(twl_store *s, int32_t pair, int32_t side, float *out, int32_t max_cols)
{
    if (!s || !s->prepared || pair < 0 || pair >= s->n_pairs || side < 0 || side > 1 || !out) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    const size_t CW = (size_t)s->P + 2;
    const int32_t n = std::min(max_cols, s->h_len[2 * (size_t)pair + side]);
    std::lock_guard<std::mutex> lk(s->d->mu);
    HIP_TRY(hipSetDevice(s->d->id));
    if (n > 0)
        HIP_TRY(hipMemcpy(out, (const float *)s->lv->d_cols.p + ((size_t)pair * 2 + side) * (size_t)s->seq_len * CW, (size_t)n * CW * sizeof(float), hipMemcpyDeviceToHost));
    return TWL_OK;
}

int twl_level_timing(twl_store *s, double *prepare_ms, double *commit_ms)
{
    if (!s) { g_err = "bad argument"; return TWL_ERR_BAD_ARGUMENT; }
    if (s->commit_pending) {
        std::lock_guard<std::mutex> lk(s->d->mu);
        HIP_TRY(hipSetDevice(s->d->id));
        HIP_TRY(hipEventSynchronize(s->d->ev[7]));
        float ms = 0.f, ms2 = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->d->ev[6], s->d->ev[7]));
        if (s->commit_side) { HIP_TRY(hipEventSynchronize(s->d->ev2[1])); HIP_TRY(hipEventElapsedTime(&ms2, s->d->ev2[0], s->d->ev2[1])); }
        s->commit_ms = ms + ms2;
        s->commit_pending = false;
    }
    if (prepare_ms) *prepare_ms = s->prepare_ms;
    if (commit_ms) *commit_ms = s->commit_ms;
    return TWL_OK;
}

}  // extern "C"

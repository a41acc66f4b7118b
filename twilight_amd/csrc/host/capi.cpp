// twilight_amd/csrc/host/capi.cpp -- C ABI of libtwl_host (include/twl_msa.h): the DEFAULT_ALN flow of driver.cpp in steps.
#include <omp.h>
#include <sched.h>
#include <thread>
#include <algorithm>
#include <cstdlib>
#include "../../../include/twl_msa.h"
#include "../../../include/twl_align.h"

#include "twl_host.hpp"

#include <chrono>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#ifndef TWL_BUILD_STAMP
#define TWL_BUILD_STAMP "unstamped"
#endif
// digest of this artefact's sources, headers and flags (__graft_entry__.build() rebuilds when the file does not carry the current one)
__attribute__((used)) static const char twl_build_stamp[] = "TWLSTAMP:" TWL_BUILD_STAMP ";";

namespace {
thread_local std::string g_msaErr;
double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

struct twl_msa {
    msa::Option option;
    msa::Params *param = nullptr;
    msa::SequenceDB *db = nullptr;
    msa::Tree *T = nullptr, *subT = nullptr;
    bool hostStaged = false, aligned = false;
    int alnLen = 0;
    double alignS = 0;
};

extern "C" {

const char *twl_msa_last_error(void) { return g_msaErr.c_str(); }

int twl_msa_open(int argc, const char *const *argv, twl_msa **out)
{
    if (!out || argc < 1 || !argv) { g_msaErr = "bad argument"; return -2; }
    auto *m = new twl_msa();
    std::vector<std::string> keep(argv, argv + argc);
    std::vector<char *> av;
    for (auto &s : keep) av.push_back(const_cast<char *>(s.c_str()));
    if (!msa::parseCommandLine((int)av.size(), av.data(), m->option)) { delete m; g_msaErr = "missing -t / -i / -o"; return -2; }
    m->hostStaged = m->option.hostStaged;
    msa::progressive::gpu::beginInit(&m->option);
    m->db = new msa::SequenceDB();
    m->db->updateSeqTh = m->option.updateSeqTh;
    m->db->lazyRows = true;        // the rows stay in HBM after the main pass until twl_msa_write (or a deferred pass) needs them
    m->param = new msa::Params(m->option, m->option.type);
    m->T = new msa::Tree(m->option.treeFile);                                     // twilight-main.cpp:122
    phylogeny::assignSinglePartition(m->T->root);                                 // :129-130 with maxSubtree = INT32_MAX
    m->subT = new msa::Tree(m->T->root, m->option.reroot);                        // :145
    msa::io::readSequences(m->option.seqFile, m->db, &m->option, m->subT);       // :146
    *out = m;
    return 0;
}

int twl_msa_shard(twl_msa *m, int rank, int world, twl_msa_exchange_fn exchange, void *user)
{
    if (!m || world < 1 || rank < 0 || rank >= world || (world > 1 && !exchange)) { g_msaErr = "bad argument"; return -2; }
    if (m->aligned) { g_msaErr = "already aligned"; return -2; }
    // Several ranks share the host: launchers export OMP_NUM_THREADS=1 for multi-process jobs (torch.distributed.run does), which would
    // leave the host part of every level (gappy columns back, path blocks) on one core.  Each rank takes its share of the cores this
    // process may run on instead; TWL_OMP_THREADS overrides.
    if (world > 1) {
        int threads = 0;
        if (const char *v = getenv("TWL_OMP_THREADS")) threads = atoi(v);
        if (threads <= 0) {
            cpu_set_t set;
            CPU_ZERO(&set);
            const int cores = (sched_getaffinity(0, sizeof set, &set) == 0) ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
            threads = std::max(1, cores / world);
        }
        omp_set_num_threads(threads);
    }
    msa::progressive::gpu::Shard sh;
    sh.rank = rank; sh.world = world; sh.exchange = exchange; sh.user = user;
    msa::progressive::gpu::setShard(m->db, sh);
    if (m->hostStaged) m->db->ownedPrefix = nullptr;      // (subtree ownership is the device-resident kernel's)
    return 0;
}

int twl_msa_shard_device(twl_msa *m, int rank, int world, twl_msa_exchange_fn exchange_dev, void *user_dev, twl_msa_exchange_fn exchange, void *user)
{
    if (!exchange_dev) { g_msaErr = "bad argument"; return -2; }
    const int rc = twl_msa_shard(m, rank, world, exchange ? exchange : exchange_dev, user);      // (validation, thread share)
    if (rc) return rc;
    msa::progressive::gpu::Shard sh;
    sh.rank = rank; sh.world = world; sh.exchange = exchange; sh.user = user; sh.exchangeDev = exchange_dev; sh.userDev = user_dev;
    msa::progressive::gpu::setShard(m->db, sh);
    if (m->hostStaged) m->db->ownedPrefix = nullptr;
    return 0;
}

int twl_msa_rccl_unique_id(void *id128)
{
    const int rc = twl_comm_unique_id(id128);
    if (rc != TWL_OK) { g_msaErr = twl_last_error(); return -1; }
    return 0;
}

int twl_msa_shard_rccl(twl_msa *m, int rank, int world, const void *id128)
{
    if (!m || !id128 || world < 1 || rank < 0 || rank >= world) { g_msaErr = "bad argument"; return -2; }
    if (m->hostStaged) { g_msaErr = "the library's own collective serves the device-resident level kernel"; return -2; }
    const int rc = twl_msa_shard(m, rank, world, [](void *, const void *, int64_t, void *) { return -1; }, nullptr);      // (validation, thread share; the callback is replaced below)
    if (rc) return rc;
    const int rcComm = msa::progressive::gpu::initRcclShard(m->db, &m->option, rank, world, id128);
    if (rcComm) {                           // no communicator: the handle is as it was before the call (the caller may shard it through its own collective)
        g_msaErr = std::string("twl_comm_init failed: ") + twl_last_error();
        twl_msa_shard(m, 0, 1, nullptr, nullptr);
        return -3;
    }
    return 0;
}

int twl_msa_upload(twl_msa *m)
{
    if (!m) { g_msaErr = "bad argument"; return -2; }
    if (!m->hostStaged && !m->aligned) msa::progressive::gpu::uploadSequences(m->db, &m->option);
    return 0;
}

int twl_msa_align(twl_msa *m)
{
    if (!m) { g_msaErr = "bad argument"; return -2; }
    if (m->aligned) { g_msaErr = "already aligned: open a fresh handle"; return -2; }
    msa::alnFunction kernel = msa::progressive::gpu::alignmentKernel_Resident;
    if (m->hostStaged) kernel = msa::progressive::gpu::alignmentKernel_GPU;
    const double t0 = nowS();
    msa::progressive::msaOnSubtree(m->subT, m->db, &m->option, *m->param, kernel, kernel);   // :148
    m->alignS = nowS() - t0;
    if (m->option.debug) msa::progressive::gpu::downloadRows(m->db, m->subT);
    if (m->option.debug && !m->db->debug()) std::cerr << "WARNING: --check found an illegal alignment row.\n";
    m->alnLen = m->subT->root->getAlnLen(m->db->currentTask);
    m->aligned = true;
    return 0;
}

int twl_msa_report(twl_msa *m, twl_msa_totals *t, twl_msa_level *levels, int32_t max_levels)
{
    if (!m || !t) { g_msaErr = "bad argument"; return -2; }
    const auto &recs = msa::progressive::gpu::levelRecords(m->db);
    const auto &tot = msa::progressive::gpu::runTotals(m->db);
    memset(t, 0, sizeof *t);
    t->n_levels = (int32_t)recs.size();
    t->aln_len = m->alnLen;
    t->n_sequences = (int32_t)m->db->sequences.size();
    t->pairs = tot.pairs; t->band_cells = tot.band_cells; t->relaunched = tot.relaunched;
    t->kernel_ms = tot.kernel_ms; t->exchange_ms = tot.exchange_ms; t->align_s = m->alignS;
    t->nominal_cells = tot.nominal_cells;
    for (int32_t i = 0; levels && i < max_levels && i < (int32_t)recs.size(); ++i) {
        levels[i].pairs = recs[i].pairs; levels[i].task = recs[i].task; levels[i].band_cells = recs[i].band_cells;
        levels[i].relaunched = recs[i].relaunched; levels[i].kernel_ms = recs[i].kernel_ms; levels[i].level_ms = recs[i].level_ms;
        levels[i].exchange_ms = recs[i].exchange_ms; levels[i].matrix_mode = recs[i].matrix_mode; levels[i].speculative = recs[i].speculative;
        levels[i].mt_tiles_predicted = recs[i].mt_predicted; levels[i].mt_tiles_inline = recs[i].mt_inline;
        memcpy(levels[i].kernel, recs[i].kernel, sizeof levels[i].kernel);
    }
    return 0;
}

int twl_msa_write(twl_msa *m, const char *path)
{
    if (!m || !m->aligned) { g_msaErr = "not aligned yet"; return -2; }
    msa::progressive::gpu::downloadRows(m->db, m->subT);
    const std::string keep = m->option.outFile;
    if (path) m->option.outFile = path;
    msa::io::writeFinalMSA(m->db, &m->option, m->alnLen);                          // :165
    m->option.outFile = keep;
    return 0;
}

void twl_msa_close(twl_msa *m)
{
    if (!m) return;
    delete m->db;
    delete m->param;
    delete m->subT;
    delete m->T;
    delete m;
}

}  // extern "C"

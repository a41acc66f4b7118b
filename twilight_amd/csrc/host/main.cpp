// twilight_amd/csrc/host/main.cpp -- `twilight-mi355x`: TWILIGHT's tree+sequences mode with the MI355X level kernel.
//   twilight-mi355x -t tree.nwk -i seqs.fa -o out.aln [-v] [--check] [--gpu-index 0,1,...] [scoring flags as in TWILIGHT]
#include "twl_host.hpp"

#include "../../../include/twl_align.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <iostream>
#include <new>
#include <thread>
#include <vector>

#include <csignal>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#ifndef TWL_BUILD_STAMP
#define TWL_BUILD_STAMP "unstamped"
#endif
// digest of this artefact's sources, headers and flags (__graft_entry__.build() rebuilds when the file does not carry the current one)
__attribute__((used)) static const char twl_build_stamp[] = "TWLSTAMP:" TWL_BUILD_STAMP ";";

// --gpu-index a,b,...: one PROCESS per GPU, forked here before anything touches the GPU.  The ranks align the same family
// together: the pairs of every level are dealt to them, each aligns its share on its device, and the final paths meet in ONE ncclAllGather
// per level, HBM to HBM over xGMI, made by the library itself (include/twl_align.h, twl_comm_*).  The 128-byte communicator id goes from
// rank 0 to the others through a page the processes share; rank 0 (this process) writes the MSA.
//
// A rank that fails must not leave the others behind: they may be spinning on the page or blocked inside ncclCommInitRank / an
// all-gather, which no return code ever leaves.  So: every rank marks the page `failed` from an exit hook unless it got to its normal end; the
// children ask the kernel for SIGTERM when their parent dies and watch the page and getppid() from a thread of their own; rank 0 reaps its children
// from a watchdog thread while the run is going on -- a child that ends badly, or a page marked failed, ends the whole run at once: the children
// are killed and the process leaves with an error.  Nothing is ever re-executed.
namespace {
struct SharedPage { std::atomic<int> ready; std::atomic<int> failed; char id[TWL_COMM_ID_BYTES]; };
SharedPage *g_page = nullptr;
std::atomic<int> g_finishedOk{0};
void markFailedAtExit() { if (g_page && !g_finishedOk.load()) g_page->failed = 1; }
void killAll(const std::vector<pid_t> &kids) { for (pid_t k : kids) if (k > 0) kill(k, SIGTERM); }
}

int main(int argc, char **argv)
{
    msa::Option option;
    if (!msa::parseCommandLine(argc, argv, option)) {
        std::cerr << "usage: twilight-mi355x -t <tree.nwk> -i <sequences.fa[.gz]> -o <out.aln> [-r 0.95] [--type n|p] [--match 18 --mismatch -8 --transition -4\n"
                     "        --gap-open -50 --gap-extend -5 --gap-ends X --xdrop 600] [-w] [--rooted] [--filter] [--check] [-v] [--gpu-index 0,1] [--host-staged]\n";
        return 1;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    int rank = 0;
    const std::vector<int> allDevices = option.gpuIdx;
    const int world = (allDevices.size() > 1 && (!option.hostStaged || option.testForkHostStaged) && option.testVirtualDevices == 0) ? (int)allDevices.size() : 1;
    SharedPage *page = nullptr;
    std::vector<pid_t> kids;
    if (world > 1) {
        void *mem = mmap(nullptr, sizeof(SharedPage), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (mem == MAP_FAILED) { std::cerr << "ERROR: mmap of the page the ranks share failed.\n"; return 1; }
        page = new (mem) SharedPage();
        page->ready = 0; page->failed = 0;
        const pid_t parent = getpid();
        for (int r = 1; r < world; ++r) {
            const pid_t pid = fork();
            if (pid < 0) { std::cerr << "ERROR: fork failed.\n"; page->failed = 1; killAll(kids); for (pid_t k : kids) waitpid(k, nullptr, 0); return 1; }
            if (pid == 0) {
                rank = r; kids.clear();
                prctl(PR_SET_PDEATHSIG, SIGTERM);
                if (getppid() != parent) _exit(1);      // (the parent was gone before the request took effect)
                break;
            }
            kids.push_back(pid);
        }
        g_page = page;
        atexit(markFailedAtExit);
        if (rank != 0) {
            // a child blocked in a collective never sees a return code: this thread ends the process when the run has failed elsewhere
            std::thread([page, parent]() {
                for (;;) {
                    if (page->failed.load() || getppid() != parent) _exit(1);
                    std::this_thread::sleep_for(std::chrono::milliseconds(20));
                }
            }).detach();
        }
        option.gpuIdx.assign(1, allDevices[rank]);
        if (rank != 0) {       // the other ranks say nothing unless asked to (TWL_DEBUG)
            option.printDetail = false;
            if (!getenv("TWL_DEBUG")) { const int nul = open("/dev/null", O_WRONLY); if (nul >= 0) { dup2(nul, 1); dup2(nul, 2); close(nul); } }
        }
    }
    // rank 0: reap the children while the run is going on
    std::thread watchdog;
    if (world > 1 && rank == 0) {
        watchdog = std::thread([&]() {
            std::vector<pid_t> left = kids;
            while (!left.empty()) {
                bool bad = page->failed.load() != 0;
                for (pid_t &k : left) {
                    int st = 0;
                    const pid_t r = waitpid(k, &st, WNOHANG);
                    if (r == 0) continue;
                    if (r < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = true;
                    k = 0;
                }
                left.erase(std::remove(left.begin(), left.end(), (pid_t)0), left.end());
                if (bad) {
                    std::cerr << "ERROR: a rank failed; ending the run.\n";
                    page->failed = 1;
                    killAll(left);
                    for (pid_t k : left) waitpid(k, nullptr, 0);
                    _exit(1);      // (this process may be inside a collective that will never return)
                }
                if (!left.empty()) std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    msa::progressive::gpu::beginInit(&option);
    // exit handlers run last-registered first: registered again behind whatever the GPU runtime registered while it came up, the mark is set
    // BEFORE the runtime's own teardown, which may block while other ranks sit in a collective
    if (world > 1) atexit(markFailedAtExit);
    // both passes run on the GPU level kernel (the reference hard-wires its CPU kernel for the deferred pass)
    msa::alnFunction kernel = msa::progressive::gpu::alignmentKernel_Resident;
    if (option.hostStaged) kernel = msa::progressive::gpu::alignmentKernel_GPU;
    msa::progressive::gpu::LevelTotals g;      // the run's totals (they live with the run's SequenceDB)
    auto beforeAlign = [&](msa::SequenceDB *db) {
        if (world == 1) return;
        if (rank == 0) {
            msa::progressive::gpu::ensureDevicesUp(&option);
            if (twl_comm_unique_id(page->id) != TWL_OK) { std::cerr << "ERROR: " << twl_last_error() << '\n'; page->failed = 1; exit(1); }
            atexit(markFailedAtExit);      // (the devices are up: behind the runtime's handlers)
            page->ready = 1;
        } else {
            while (!page->ready.load() && !page->failed.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));      // (the watcher thread covers a parent that is gone)
            if (page->failed.load()) exit(1);
        }
        const int rcComm = msa::progressive::gpu::initRcclShard(db, &option, rank, world, page->id);
        if (rcComm != TWL_OK) { std::cerr << "ERROR: twl_comm_init failed (" << rcComm << "): " << twl_last_error() << '\n'; markFailedAtExit(); exit(1); }
        atexit(markFailedAtExit);      // (... and behind RCCL's and the device runtime's handlers, see above)
        if (option.hostStaged) db->ownedPrefix = nullptr;      // (subtree ownership is the device-resident kernel's: --test-fork-host-staged deals and gathers every level)
    };
    const int alnLen = msa::runDefaultAlignment(option, kernel, kernel, rank == 0, [&](msa::SequenceDB *db) { g = msa::progressive::gpu::runTotals(db); }, beforeAlign);
    g_finishedOk = 1;
    if (rank != 0) _exit(0);
    if (watchdog.joinable()) watchdog.join();      // every child has ended well (a bad end leaves from the watchdog)
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cerr << "Wrote " << option.outFile << " (length " << alnLen << ") in " << secs << " s; level kernel: " << g.pairs << " pairs, " << g.band_cells
              << " band cells, " << g.relaunched << " pairs re-run in a wider window, " << g.kernel_ms << " ms DP kernel, " << g.total_ms << " ms incl. transfers\n";
    if (option.printDetail)
        std::cerr << "Host phases (ms): prepare " << g.prepare_ms << ", stage " << g.stage_ms << ", boundary call " << g.call_ms << ", finish " << g.finish_ms
                  << "; device prepare kernels " << g.dev_prepare_ms << ", device write-back kernels " << g.dev_commit_ms << '\n';
    return 0;
}

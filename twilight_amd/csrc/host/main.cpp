// twilight_amd/csrc/host/main.cpp -- `twilight-mi355x`: TWILIGHT's tree+sequences mode with the MI355X level kernel.
//   twilight-mi355x -t tree.nwk -i seqs.fa -o out.aln [-v] [--check] [--gpu-index 0,1,...] [scoring flags as in TWILIGHT]
#include "twl_host.hpp"

#include "../../../include/twl_align.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <iostream>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

// --gpu-index a,b,...: one PROCESS per GPU, forked here before anything touches the GPU.  The ranks align the same family
// together: the pairs of every level are dealt to them, each aligns its share on its device, and the final paths meet in ONE ncclAllGather
// per level, HBM to HBM over xGMI, made by the library itself (include/twl_align.h, twl_comm_*).  The 128-byte communicator id goes from
// rank 0 to the others through a page the processes share; rank 0 (this process) writes the MSA.
namespace {
struct SharedPage { std::atomic<int> ready; std::atomic<int> failed; char id[TWL_COMM_ID_BYTES]; };
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
    const int world = (allDevices.size() > 1 && !option.hostStaged && option.testVirtualDevices == 0) ? (int)allDevices.size() : 1;
    SharedPage *page = nullptr;
    std::vector<pid_t> kids;
    if (world > 1) {
        void *mem = mmap(nullptr, sizeof(SharedPage), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (mem == MAP_FAILED) { std::cerr << "ERROR: mmap of the page the ranks share failed.\n"; return 1; }
        page = new (mem) SharedPage();
        page->ready = 0; page->failed = 0;
        for (int r = 1; r < world; ++r) {
            const pid_t pid = fork();
            if (pid < 0) { std::cerr << "ERROR: fork failed.\n"; return 1; }
            if (pid == 0) { rank = r; kids.clear(); break; }
            kids.push_back(pid);
        }
        option.gpuIdx.assign(1, allDevices[rank]);
        if (rank != 0) {       // the other ranks say nothing unless asked to (TWL_DEBUG)
            option.printDetail = false;
            if (!getenv("TWL_DEBUG")) { const int nul = open("/dev/null", O_WRONLY); if (nul >= 0) { dup2(nul, 1); dup2(nul, 2); close(nul); } }
        }
    }
    msa::progressive::gpu::beginInit(&option);
    // both passes run on the GPU level kernel (the reference hard-wires its CPU kernel for the deferred pass)
    msa::alnFunction kernel = msa::progressive::gpu::alignmentKernel_Resident;
    if (option.hostStaged) kernel = msa::progressive::gpu::alignmentKernel_GPU;
    msa::progressive::gpu::LevelTotals g;      // the run's totals (they live with the run's SequenceDB)
    auto beforeAlign = [&](msa::SequenceDB *db) {
        if (world == 1) return;
        if (rank == 0) {
            msa::progressive::gpu::ensureDevicesUp(&option);
            if (twl_comm_unique_id(page->id) != TWL_OK) { std::cerr << "ERROR: " << twl_last_error() << '\n'; page->failed = 1; exit(1); }
            page->ready = 1;
        } else {
            while (!page->ready.load() && !page->failed.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            if (page->failed.load()) exit(1);
        }
        const int rcComm = msa::progressive::gpu::initRcclShard(db, &option, rank, world, page->id);
        if (rcComm != TWL_OK) { std::cerr << "ERROR: twl_comm_init failed (" << rcComm << "): " << twl_last_error() << '\n'; exit(1); }
    };
    const int alnLen = msa::runDefaultAlignment(option, kernel, kernel, rank == 0, [&](msa::SequenceDB *db) { g = msa::progressive::gpu::runTotals(db); }, beforeAlign);
    if (rank != 0) _exit(0);
    int bad = 0;
    for (pid_t k : kids) { int st = 0; if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad; }
    if (bad) { std::cerr << "ERROR: " << bad << " rank(s) failed.\n"; return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cerr << "Wrote " << option.outFile << " (length " << alnLen << ") in " << secs << " s; level kernel: " << g.pairs << " pairs, " << g.band_cells
              << " band cells, " << g.relaunched << " pairs re-run in a wider window, " << g.kernel_ms << " ms DP kernel, " << g.total_ms << " ms incl. transfers\n";
    if (option.printDetail)
        std::cerr << "Host phases (ms): prepare " << g.prepare_ms << ", stage " << g.stage_ms << ", boundary call " << g.call_ms << ", finish " << g.finish_ms
                  << "; device prepare kernels " << g.dev_prepare_ms << ", device write-back kernels " << g.dev_commit_ms << '\n';
    return 0;
}

// twilight_amd/csrc/host/main.cpp -- `twilight-mi355x`: TWILIGHT's tree+sequences mode with the MI355X level kernel.
//   twilight-mi355x -t tree.nwk -i seqs.fa -o out.aln [-v] [--check] [--gpu-index 0,1,...] [scoring flags as in TWILIGHT]
#include "twl_host.hpp"

#include <chrono>
#include <iostream>

int main(int argc, char **argv)
{
    msa::Option option;
    if (!msa::parseCommandLine(argc, argv, option)) {
        std::cerr << "usage: twilight-mi355x -t <tree.nwk> -i <sequences.fa[.gz]> -o <out.aln> [-r 0.95] [--type n|p] [--match 18 --mismatch -8 --transition -4\n"
                     "        --gap-open -50 --gap-extend -5 --gap-ends X --xdrop 600] [-w] [--rooted] [--filter] [--check] [-v] [--gpu-index 0,1] [--host-staged]\n";
        return 1;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    msa::progressive::gpu::beginInit(&option);
    // both passes run on the GPU level kernel (the reference hard-wires its CPU kernel for the deferred pass)
    msa::alnFunction kernel = msa::progressive::gpu::alignmentKernel_Resident;
    if (option.hostStaged) kernel = msa::progressive::gpu::alignmentKernel_GPU;
    msa::progressive::gpu::LevelTotals g;      // the run's totals (they live with the run's SequenceDB)
    const int alnLen = msa::runDefaultAlignment(option, kernel, kernel, true, [&](msa::SequenceDB *db) { g = msa::progressive::gpu::runTotals(db); });
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cerr << "Wrote " << option.outFile << " (length " << alnLen << ") in " << secs << " s; level kernel: " << g.pairs << " pairs, " << g.band_cells
              << " band cells, " << g.relaunched << " pairs re-run in a wider window, " << g.kernel_ms << " ms DP kernel, " << g.total_ms << " ms incl. transfers\n";
    if (option.printDetail)
        std::cerr << "Host phases (ms): prepare " << g.prepare_ms << ", stage " << g.stage_ms << ", boundary call " << g.call_ms << ", finish " << g.finish_ms
                  << "; device prepare kernels " << g.dev_prepare_ms << ", device write-back kernels " << g.dev_commit_ms << '\n';
    return 0;
}

// twilight_amd/csrc/host/driver.cpp -- DEFAULT_ALN flow of /root/reference/src/twilight-main.cpp:115-176 for the
// single-partition case (no -m), plus the handful of CLI flags the hot path needs (names as in twilight-main.cpp:13-84).
#include "twl_host.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <thread>

#include <omp.h>

namespace msa {

// CPUs this process may really use: the cgroup quota (cpu.max) can be far below what the OS reports
static int effectiveCpus()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string quota;
    long period = 0;
    if (f >> quota >> period && quota != "max" && period > 0) {
        long q = atol(quota.c_str());
        if (q > 0) n = std::min<long>(n, std::max<long>(1, (q + period - 1) / period));
    }
    return n;
}

static bool flag(const char *a, const char *s, const char *l) { return (s && !strcmp(a, s)) || (l && !strcmp(a, l)); }

bool parseCommandLine(int argc, char **argv, Option &o)
{
    bool typeGiven = false;
    o.cpuNum = 0;
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) { std::cerr << "ERROR: missing value for " << a << '\n'; exit(1); } return argv[++i]; };
        if (flag(a, "-t", "--tree")) o.treeFile = val();
        else if (flag(a, "-i", "--sequences")) o.seqFile = val();
        else if (flag(a, "-o", "--output")) o.outFile = val();
        else if (flag(a, "-r", "--remove-gappy")) o.gappyVertical = (float)atof(val());
        else if (flag(a, "-C", "--cpu")) o.cpuNum = atoi(val());
        else if (flag(a, "-G", "--gpu")) o.gpuNum = atoi(val());
        else if (flag(a, nullptr, "--gpu-index")) { o.gpuIdx.clear(); for (char *tok = strtok(const_cast<char *>(val()), ","); tok; tok = strtok(nullptr, ",")) o.gpuIdx.push_back(atoi(tok)); }
        else if (flag(a, nullptr, "--cpu-only")) o.cpuOnly = true;
        else if (flag(a, nullptr, "--type")) { o.type = val()[0]; typeGiven = true; }
        else if (flag(a, "-w", "--wildcard")) o.wildcard = true;
        else if (flag(a, nullptr, "--rooted")) o.reroot = false;
        else if (flag(a, nullptr, "--match")) o.match = (float)atof(val());
        else if (flag(a, nullptr, "--mismatch")) o.mismatch = (float)atof(val());
        else if (flag(a, nullptr, "--transition")) o.transition = (float)atof(val());
        else if (flag(a, nullptr, "--gap-open")) o.gapOpen = (float)atof(val());
        else if (flag(a, nullptr, "--gap-extend")) o.gapExtend = (float)atof(val());
        else if (flag(a, nullptr, "--gap-ends")) { o.gapEnds = (float)atof(val()); o.hasGapEnds = true; }
        else if (flag(a, nullptr, "--xdrop")) o.xdrop = (float)atof(val());
        else if (flag(a, "-b", "--blosum")) o.blosum = atoi(val());
        else if (flag(a, "-x", "--matrix")) o.matrixFile = val();
        else if (flag(a, nullptr, "--length-deviation")) o.lenDev = (float)atof(val());
        else if (flag(a, nullptr, "--max-ambig")) o.maxAmbig = (float)atof(val());
        else if (flag(a, nullptr, "--max-len")) o.maxLen = atoi(val());
        else if (flag(a, nullptr, "--min-len")) o.minLen = atoi(val());
        else if (flag(a, nullptr, "--filter")) o.noFilter = false;
        else if (flag(a, nullptr, "--check")) o.debug = true;
        else if (flag(a, "-v", "--verbose")) o.printDetail = true;
        else if (flag(a, "--host-staged", "--host-staged")) o.hostStaged = true;
        // development / test switches (documented in include/twl_msa.h): the reference's two thresholds, replicas of the store on one device
        else if (flag(a, nullptr, "--test-cal-profile-th")) { const int v = atoi(val()); if (v > 0) o.calProfileTh = v; }
        else if (flag(a, nullptr, "--test-update-seq-th")) { const int v = atoi(val()); if (v > 0) o.updateSeqTh = v; }
        else if (flag(a, nullptr, "--test-virtual-devices")) o.testVirtualDevices = std::max(0, atoi(val()));
        else if (flag(a, nullptr, "--test-no-ownership")) o.testNoOwnership = true;
        else if (flag(a, nullptr, "--test-fork-host-staged")) o.testForkHostStaged = true;
        else if (flag(a, nullptr, "--overwrite")) {}
        else if (flag(a, "-h", "--help")) return false;
        else { std::cerr << "ERROR: unsupported option " << a << " (this build covers the tree+sequences alignment mode only)\n"; exit(1); }
    }
    if (o.treeFile.empty() || o.seqFile.empty() || o.outFile.empty()) return false;
    if (o.cpuOnly) {       // the reference's GPU builds route to their CPU kernel here (hip/alignment-gpu.hip.cpp:19-21); this build has no CPU alignment path
        std::cerr << "ERROR: --cpu-only is not available: twilight-mi355x has no CPU alignment path (the CPU checker oracle/e2e_oracle is test infrastructure).\n";
        exit(1);
    }
    if (o.gappyVertical > 1 || o.gappyVertical <= 0) { std::cerr << "ERROR: Invalid value for --remove-gappy. The value of --remove-gappy should be in (0,1]\n"; exit(1); }
    if (o.gpuIdx.empty() && o.gpuNum > 0) for (int g = 0; g < o.gpuNum; ++g) o.gpuIdx.push_back(g);
    if (!typeGiven) o.type = io::detectType(o.seqFile);
    const int maxCpu = effectiveCpus();
    if (o.cpuNum <= 0 || o.cpuNum > maxCpu) o.cpuNum = maxCpu;       // -C/--cpu, default: all usable cores (option.cpp:41-46)
    omp_set_num_threads(o.cpuNum);
    return true;
}

int runDefaultAlignment(Option &option, alnFunction kernel, alnFunction deferredKernel, bool writeOutput, const std::function<void(SequenceDB *)> &atEnd,
                        const std::function<void(SequenceDB *)> &beforeAlign)
{
    auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = clk();
    SequenceDB database;
    database.updateSeqTh = option.updateSeqTh;
    Params param(option, option.type);
    Tree *T = new Tree(option.treeFile);                                    // twilight-main.cpp:122
    phylogeny::assignSinglePartition(T->root);                              // :129-130 with maxSubtree = INT32_MAX
    Tree *subT = new Tree(T->root, option.reroot);                          // :145
    const double t1 = clk();
    io::readSequences(option.seqFile, &database, &option, subT);           // :146
    const double t2 = clk();
    if (beforeAlign) beforeAlign(&database);      // (a sharded run sets its communicator up here: main.cpp)
    progressive::msaOnSubtree(subT, &database, &option, param, kernel, deferredKernel);   // :148
    const double t3 = clk();
    if (option.debug && !database.debug()) std::cerr << "WARNING: --check found an illegal alignment row.\n";
    const int alnLen = subT->root->getAlnLen(database.currentTask);
    if (writeOutput) io::writeFinalMSA(&database, &option, alnLen);        // :165
    if (option.printDetail)
        std::cerr << "Driver phases (s): tree " << t1 - t0 << ", read " << t2 - t1 << ", align " << t3 - t2 << ", write " << clk() - t3 << '\n';
    if (atEnd) atEnd(&database);
    delete subT;
    delete T;
    return alnLen;
}

}  // namespace msa

// twilight_amd/csrc/host/twl_host.hpp -- host-side mirror of TWILIGHT's level-batch call surface.
//
// Same namespaces, type and function names, argument meaning and error behaviour as the reference
// (/root/reference/src/msa.hpp, phylogeny.hpp) for the DEFAULT_ALN path, so that the level kernel
// msa::progressive::gpu::alignmentKernel_GPU (align_gpu.cpp, calls the C ABI of include/twl_align.h)
// is injected exactly where the reference injects cpu::alignmentKernel_CPU (twilight-main.cpp:148).
// No Boost, no TBB: options are a plain struct, parallel loops are OpenMP.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <functional>
#include <map>
#include <stack>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace phylogeny {

using Profile = std::vector<std::vector<float>>;

// reference phylogeny.hpp:16-53
struct Node {
    Node(const std::string &id, float len);
    Node(const std::string &id, Node *par, float len);
    bool is_leaf() const { return identifier.compare(0, 4, "node") != 0; }
    void collectPostOrder(std::stack<Node *> &postStack);

    std::string identifier;
    Node *parent = nullptr;
    float branchLength = 0;
    size_t level = 1;
    std::vector<Node *> children;
    size_t numLeaves = 0;
    float weight = 0;
    bool placed = false;
    int grpID = -1;
    int schedIdx = -1;         // scratch of progressive::getProgressivePairs (position in its post-order walk)

    std::vector<int> seqsIncluded;
    Profile msaFreq;
    int cacheId = -1;          // device-resident mode: handle of the profile cached in the store (plays msaFreq; -1 = none)
    int alnLen = 0;
    int alnNum = 0;
    float alnWeight = 0;
    int getAlnNum(int) const { return alnNum; }
    int getAlnLen(int) const { return alnLen; }
};

// reference phylogeny.hpp:73-107 (the members the DEFAULT_ALN path touches)
struct Tree {
    size_t m_currInternalNode = 0;
    size_t m_maxDepth = 0;
    size_t m_numLeaves = 0;
    float m_meanDepth = 0;
    Node *root = nullptr;
    std::unordered_map<std::string, Node *> allNodes;   // iteration order is observable (reroot start leaf, tree.cpp:601-605)

    std::string newInternalNodeId() { return "node_" + std::to_string(++m_currInternalNode); }
    void calLeafNum();
    void calSeqWeight();
    void parseNewick(std::string &newick);
    void reroot();
    void convert2binaryTree();
    Tree *prune(std::unordered_set<std::string> &seqs);

    Tree() = default;
    explicit Tree(const std::string &newickFileName);
    Tree(Node *node, bool reroot);          // copy of the nodes that share node->grpID
    ~Tree();
};

void pruneTree(Tree *&T, std::unordered_set<std::string> &seqs);
void updateLevels(Node *node, size_t currentLevel);
// single-partition form of PartitionInfo::partitionTree (partitionInfo.cpp:75-85): every node joins group 0
void assignSinglePartition(Node *root);

}  // namespace phylogeny

char checkOnly(char c);
int letterIdx(char type, char c);

namespace msa {

enum Type { DEFAULT_ALN = 0, MERGE_MSA = 1, PLACE_WO_TREE = 2, PLACE_W_TREE = 3 };

using Node = phylogeny::Node;
using Tree = phylogeny::Tree;
using NodePair = std::pair<Node *, Node *>;
using NodePairVec = std::vector<NodePair>;
using stringPair = std::pair<std::string, std::string>;
using IntPair = std::pair<int, int>;
using IntPairVec = std::vector<IntPair>;
using FloatPair = std::pair<float, float>;
using alnPath = std::vector<int8_t>;
using Profile = std::vector<std::vector<float>>;

// reference msa.hpp:55-96; values are the CLI defaults of twilight-main.cpp:13-84
struct Option {
    int alnMode = DEFAULT_ALN;
    int gpuNum = 0;
    int cpuNum = 1;
    std::vector<int> gpuIdx;
    bool cpuOnly = false;
    int maxSubtree = INT32_MAX;
    float gappyVertical = 0.95f;
    float lenDev = 0;
    float maxAmbig = 0.1f;
    int maxLen = INT32_MAX;
    int minLen = 0;
    bool writeFiltered = false;
    bool debug = false;          // --check
    bool noFilter = true;        // !--filter
    bool reroot = true;          // !--rooted
    bool compressed = false;
    char type = 'n';
    bool alignGappy = true;
    std::string treeFile, seqFile, outFile;
    bool printDetail = false;    // -v
    // msa.hpp:179-180 (_CAL_PROFILE_TH, _UPDATE_SEQ_TH): 1000 in the reference; tests lower them per RUN (--test-cal-profile-th / --test-update-seq-th: small trees
    // then reach the cached-profile and compressed-group branches).  Per run since round 5: as process globals a handle with lowered thresholds changed every later one
    int calProfileTh = 1000, updateSeqTh = 1000;
    bool testForkHostStaged = false; // --test-fork-host-staged: the CLI forks one process per listed device for the host-staged kernel too (paths all-gathered through the library's communicator per level); lets the forked flow run on the CPU-check build
    bool testNoOwnership = false; // --test-no-ownership: a sharded run deals and exchanges every level (no subtree ownership below a cut)
    int testVirtualDevices = 0;  // --test-virtual-devices n: n replicas of the store on the first device (the several-replica path of the resident kernel on a one-GPU box)
    bool hostStaged = false;     // --host-staged: build profiles on the host and stage them per level (default: device-resident rows)
    // scoring flags (consumed by Params)
    float match = 18, mismatch = -8, transition = -4, gapOpen = -50, gapExtend = -5, xdrop = 600;
    bool hasGapEnds = false;
    float gapEnds = 0;
    bool wildcard = false;
    int blosum = 62;
    std::string matrixFile;      // -x / --matrix: user-defined substitution matrix (scoring-matrix.cpp:137-199)
};

// reference msa.hpp:98-109, ctor scoring-matrix.cpp:81-199
struct Params {
    float gapOpen, gapExtend, gapBoundary, xdrop, scaleFactor = 1;
    float **scoringMatrix;
    int matrixSize;
    Params(const Option &opt, char type);
    ~Params();
    Params(const Params &) = delete;
};

// reference msa.hpp:111-155
struct SequenceDB {
    struct SequenceInfo {
        int id;
        std::string name;
        std::string unalignedSeq;
        int len;
        bool lowQuality = false;
        int subtreeIdx;
        float weight;
        bool storage = false;
        bool borrowed = false;      // alnStorage points into SequenceDB::rowArena (device-resident mode hands all rows back in one block)
        int memLen = 0;
        char *alnStorage[2] = {nullptr, nullptr};
        static constexpr int timesBigger = 2;
        void changeStorage() { storage = !storage; }
        void memCheck(int len);
        SequenceInfo(int id_, const std::string &name_, std::string &seq, int subtreeIdx_, float weight_, bool debug);
        ~SequenceInfo();
    };
    int currentTask = 0;
    int updateSeqTh = 1000;                        // Option::updateSeqTh of the run (updateAlignment has no Option)
    std::vector<SequenceInfo *> sequences;
    std::vector<Node *> fallback_nodes;
    std::unordered_map<std::string, SequenceInfo *> name_map;
    std::unordered_map<int, alnPath> subtreeAln;
    char *rowArena = nullptr;                      // owns the rows of sequences with `borrowed` set
    void *gpuCtx = nullptr;                        // per-run state of the GPU level kernels (progressive::gpu::RunCtx), freed through gpuCtxFree
    void (*gpuCtxFree)(void *) = nullptr;
    // set for a sharded run on the device-resident level kernel: aligns a prefix of the main pass's levels by subtree ownership (align_owned.cpp), returns how many
    std::function<size_t(Tree *, std::vector<std::vector<std::pair<Node *, Node *>>> &, Option *, Params &)> ownedPrefix;
    std::function<void(Tree *)> afterMainPass;    // set by the device-resident level kernel: bring rows/caches back to the host
    bool residentDeferred = false;                 // ... and that kernel runs the deferred pass on the resident rows too: they come back after it
    bool lazyRows = false;                         // library use (twl_msa.h): leave the rows in HBM after the main pass until somebody needs them
    void addSequence(int id, const std::string &name, std::string &seq, int subtreeIdx, float weight, bool debug);
    bool debug();      // --check: true when every aligned row reproduces its input sequence and all rows are equally long
    ~SequenceDB();
};

namespace io {
void readSequences(const std::string &fileName, SequenceDB *database, Option *option, Tree *&T);
void writeAlignment(const std::string &fileName, SequenceDB *database, int alnLen);
void writeFinalMSA(SequenceDB *database, Option *option, int alnLen);
char detectType(const std::string &seqFile);
}  // namespace io

using alnFunction = std::function<void(Tree *, NodePairVec &, SequenceDB *, Option *, Params &)>;

namespace alignment_helper {
void calculateProfile(float *profile, NodePair &nodes, SequenceDB *database, Option *option, int32_t memLen);
void removeGappyColumns(float *hostFreq, NodePair &nodes, Option *option, std::pair<IntPairVec, IntPairVec> &gappyColumns, int32_t memLen,
                        IntPair &lens, int currentTask);
void calculatePSGP(float *hostFreq, float *hostGapOp, float *hostGapEx, NodePair &nodes, SequenceDB *database, Option *option, int memLen,
                   IntPair offset, IntPair lens, Params &param);
void getConsensus(Option *option, float *profile, std::string &consensus, int len);
void pairwiseGlobal(const std::string &seq1, const std::string &seq2, alnPath &path, Params &param);
void addGappyColumnsBack(alnPath &aln_before, alnPath &aln_after, std::pair<IntPairVec, IntPairVec> &gappyColumns, Params &param,
                         IntPair rgcLens, stringPair orgSeqs);
void updateAlignment(NodePair &nodes, SequenceDB *database, Option *option, alnPath &aln);
void updateFrequency(NodePair &nodes, SequenceDB *database, alnPath &aln, FloatPair weights);
void fallback2cpu(std::vector<int> &fallbackPairs, NodePairVec &nodes, SequenceDB *database, Option *option);
}  // namespace alignment_helper

namespace progressive {
void getProgressivePairs(std::vector<std::pair<NodePair, int>> &alnOrder, std::stack<Node *> postStack, int grpID, int mode);
void scheduling(Node *root, std::vector<NodePairVec> &levels, int mode);
void updateNode(Tree *tree, NodePairVec &nodes, SequenceDB *database);
void progressiveAlignment(Tree *T, SequenceDB *database, Option *option, std::vector<NodePairVec> &levels, Params &param, alnFunction kernel);
// Subtree ownership of a sharded run: levels [0, cut] are aligned by the owner of each pair's subtree alone (owner[level][pair]), the rest dealt per level.
struct OwnershipPlan { int cut = -1; int subtrees = 0; std::vector<std::vector<int>> owner; std::vector<long long> load; };
OwnershipPlan planOwnership(Tree *T, const std::vector<NodePairVec> &levels, int world);
// `deferredKernel` aligns the deferred sequences against the root in the second pass; the reference hard-wires
// cpu::alignmentKernel_CPU there (progressive.cpp:291).
void msaOnSubtree(Tree *T, SequenceDB *database, Option *option, Params &param, alnFunction kernel, alnFunction deferredKernel);
void updateAlignment(Node *node, SequenceDB *database);

// What one pair needs before / after the DP (alignment-cpu.cpp:50-93 and :136-175), shared by every level kernel.
// A float buffer that is either owned (per-pair vector) or a slot of the level's flat staging arrays (align_gpu.cpp).
struct FloatBuf {
    std::vector<float> own;
    float *ext = nullptr;
    float *data() { return ext ? ext : own.data(); }
    const float *data() const { return ext ? ext : own.data(); }
    void assign(size_t n, float v) { ext = nullptr; own.assign(n, v); }
    void bind(float *slot, size_t n) { ext = slot; std::fill(slot, slot + n, 0.0f); }
};
struct PairInputs {
    FloatBuf freq, gapOp, gapEx;                     // freq[2][memLen][P], gapOp/gapEx[2][memLen]
    std::pair<IntPairVec, IntPairVec> gappyColumns;
    stringPair consensus;
    IntPair lens;                                    // after gappy-column removal
    int32_t refLen, qryLen, refNum, qryNum, memLen;
    bool lowQ_r, lowQ_q;
};
// With slots given, the pair's buffers are built in place there with row stride `stride` (>= max(refLen, qryLen)); the stride
// is only an address stride (alignment-cpu.cpp:13-30 uses max(refLen, qryLen)), it does not enter any value.
void preparePair(NodePair &nodes, SequenceDB *database, Option *option, Params &param, PairInputs &in, float *freqSlot = nullptr,
                 float *gapOpSlot = nullptr, float *gapExSlot = nullptr, int stride = 0);
// Returns false when the pair must be deferred (fallbackPairs), true when it was written back.
bool finishPair(NodePair &nodes, SequenceDB *database, Option *option, Params &param, PairInputs &in, alnPath &aln_wo_gc);

namespace gpu {
void alignmentKernel_GPU(Tree *T, NodePairVec &alnPairs, SequenceDB *database, Option *option, Params &param);
void beginInit(Option *option);   // optional: start device initialisation early, on a helper thread
// Device-resident variant (include/twl_level.h): rows and cached profiles stay in HBM during the main progressive pass; profile
// building, gappy-column removal, gap penalties and the row write-back run as kernels.  Falls through to alignmentKernel_GPU
// for the deferred pass (currentTask != 0), after bringing the rows back.
void alignmentKernel_Resident(Tree *T, NodePairVec &alnPairs, SequenceDB *database, Option *option, Params &param);
struct LevelTotals { uint64_t band_cells = 0, pairs = 0, relaunched = 0, nominal_cells = 0 /* sum of R*Q over the pairs this process's level calls saw (all of them on one GPU) */; double kernel_ms = 0, total_ms = 0, prepare_ms = 0, stage_ms = 0, call_ms = 0, finish_ms = 0, dev_prepare_ms = 0, dev_commit_ms = 0, exchange_ms = 0; };
// One level-kernel call, as the reference's per-level report line (progressive.cpp:178-189) plus what the DP did in it.
struct LevelRecord { int32_t pairs = 0, task = 0; uint64_t band_cells = 0, relaunched = 0; double kernel_ms = 0, level_ms = 0, exchange_ms = 0; int32_t matrix_mode = -1, speculative = 0;
                     int32_t mt_predicted = 0, mt_inline = 0; char kernel[160] = {0}; };
// Several processes (one per GPU) aligning ONE family together: every process runs the same host flow on its own replica, aligns
// the pairs dealt to its rank and receives the other ranks' paths through `exchange` (an all-gather of equal-sized host blocks:
// send = this rank's block, recv = [world][bytes_per_rank]; returns 0 on success).  twilight_amd/dist.py provides it over
// torch.distributed (backend nccl = RCCL over xGMI on GPUs, gloo in the CPU tests).
using ExchangeFn = int (*)(void *user, const void *send, int64_t bytes_per_rank, void *recv);
// exchangeDev: the same all-gather on DEVICE buffers (this process's GPU); with it the device-resident level kernel keeps the paths in
// HBM end to end (align -> gappy columns back -> pack -> all-gather -> unpack -> write-back)
struct Shard { int rank = 0, world = 1; ExchangeFn exchange = nullptr; void *user = nullptr; ExchangeFn exchangeDev = nullptr; void *userDev = nullptr;
               bool rccl = false; };      // rccl: the all-gathers are the library's own (twl_comm_all_gather*, RCCL from C++), no callback
// Communicator of a sharded run on this process's device (twl_comm_init); the two all-gathers the level kernels use when Shard::rccl is set.
int initRcclShard(SequenceDB *database, Option *option, int rank, int world, const void *id128);      // 0, or the code of twl_comm_init
void ensureDevicesUp(Option *option);      // joins the twl_init started by beginInit
// Subtree ownership of a sharded run (align_owned.cpp): levels [0, returned) of the main pass are done when it returns.
size_t ownedPrefix(Tree *T, std::vector<NodePairVec> &levels, SequenceDB *database, Option *option, Params &param);
// Per-run state of the level kernels; hangs off SequenceDB::gpuCtx so that several runs can live in one process.
struct RunCtx;
RunCtx &ctxOf(SequenceDB *database);
void setShard(SequenceDB *database, const Shard &shard);
const std::vector<LevelRecord> &levelRecords(SequenceDB *database);
const LevelTotals &runTotals(SequenceDB *database);
// Device-resident mode: create the store (sequences into HBM) ahead of the first level, e.g. before a timed region.
void uploadSequences(SequenceDB *database, Option *option);
// Rows (and the root's cached profile) back to the host, if they are still in HBM.
void downloadRows(SequenceDB *database, Tree *T);
}

}  // namespace progressive

// DEFAULT_ALN driver shared by the product CLI and the oracle's end-to-end checker (twilight-main.cpp:121-176,
// single partition): tree -> partition -> reroot -> read sequences -> msaOnSubtree -> write MSA.  Returns the MSA length.
// atEnd (optional) sees the run's SequenceDB after the output was written and before it is destroyed (the CLI reads the run's totals there)
int runDefaultAlignment(Option &option, alnFunction kernel, alnFunction deferredKernel, bool writeOutput = true,
                        const std::function<void(SequenceDB *)> &atEnd = nullptr, const std::function<void(SequenceDB *)> &beforeAlign = nullptr);
bool parseCommandLine(int argc, char **argv, Option &option);

}  // namespace msa

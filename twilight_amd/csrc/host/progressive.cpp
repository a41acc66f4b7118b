// twilight_amd/csrc/host/progressive.cpp -- level schedule and the caller of the level kernel.
// Behavioural mirror of /root/reference/src/progressive.cpp:10-299 (modes 0 and 1; DEFAULT_ALN).
#include "twl_host.hpp"

#include <algorithm>
#include <chrono>
#include <iostream>
#include <unordered_map>

namespace msa {
namespace progressive {

// progressive.cpp:10-107.  mode 0: siblings are paired left to right, a pair's level is one more than the later of
// its two members' previous levels; mode 1: every node is paired with its parent (deferred pass).
void getProgressivePairs(std::vector<std::pair<NodePair, int>> &alnOrder, std::stack<Node *> postStack, int grpID, int mode)
{
    // (the reference keys a std::map by the node's identifier string; identifiers are unique within a tree -- Tree::allNodes is keyed by
    // them too -- and the map is only ever looked up, so a table indexed by the node's position in the walk serves: 100 000-leaf trees
    // spent 0.18 s in the string map, 40 ms in a pointer hash)
    struct LevelOf {          // levelOf[node], -1 = not seen yet; Node::schedIdx is the node's slot (checked against its owner: stale from an earlier call otherwise)
        std::vector<int> lvl;
        std::vector<const Node *> own;
        int &operator[](Node *n)
        {
            const int k = n->schedIdx;
            if (k < 0 || k >= (int)lvl.size() || own[k] != n) { n->schedIdx = (int)lvl.size(); lvl.push_back(-1); own.push_back(n); }
            return lvl[n->schedIdx];
        }
    } levelOf;
    levelOf.lvl.reserve(postStack.size() + 16);
    levelOf.own.reserve(postStack.size() + 16);
    auto nextLevel = [&](Node *id) { const int l = levelOf[id]; return l < 0 ? 0 : l + 1; };
    if (mode == 0) {
        std::vector<Node *> children, left;
        for (; !postStack.empty(); postStack.pop()) {
            Node *node = postStack.top();
            if (!(node->grpID == -1 || node->grpID == grpID) || node->is_leaf()) continue;
            children.clear();
            for (Node *c : node->children)
                if (c->grpID == grpID) children.push_back(c);
            if (children.empty() && node->seqsIncluded.empty()) {          // useless node: drop it from the subtree
                node->grpID = -2;
                auto &sib = node->parent->children;
                for (auto it = sib.begin(); it != sib.end(); ++it)
                    if ((*it)->identifier == node->identifier) { sib.erase(it); break; }
                continue;
            }
            if (children.size() == 1 && node->parent != nullptr && node->seqsIncluded.empty() && node->parent->grpID == grpID) {
                for (auto &slot : node->parent->children)                  // unary node: splice its child into the parent
                    if (slot->identifier == node->identifier) {
                        slot = children[0];
                        children[0]->branchLength += node->branchLength;
                        children[0]->parent = node->parent;
                        break;
                    }
                continue;
            }
            while (children.size() > 1) {
                left.clear();
                for (size_t i = 0; i + 1 < children.size(); i += 2) {
                    const int lvl = std::max(nextLevel(children[i]), nextLevel(children[i + 1]));
                    levelOf[children[i]] = lvl;
                    levelOf[children[i + 1]] = lvl;
                    alnOrder.push_back({{children[i], children[i + 1]}, lvl});
                    left.push_back(children[i]);
                }
                if (children.size() % 2 == 1) left.push_back(children.back());
                children.swap(left);
            }
            if (children.size() == 1 && !node->seqsIncluded.empty()) {
                const int lvl = std::max(nextLevel(node), nextLevel(node->children[0]));
                levelOf[node] = lvl;
                levelOf[node->children[0]] = lvl;
                alnOrder.push_back({{node, node->children[0]}, lvl});
            }
            { const int l = levelOf[children[0]]; levelOf[node] = l; }      // (operator[] may grow the table: the value first, then the slot)
        }
    } else if (mode == 1) {
        for (; !postStack.empty(); postStack.pop()) {
            Node *node = postStack.top();
            if (node->parent == nullptr) continue;
            const int lvl = std::max(nextLevel(node), nextLevel(node->parent));
            levelOf[node] = lvl;
            levelOf[node->parent] = lvl;
            alnOrder.push_back({{node->parent, node}, lvl});
        }
    } else {
        for (; !postStack.empty(); postStack.pop())
            if (postStack.top()->parent != nullptr) alnOrder.push_back({{postStack.top()->parent, postStack.top()}, 0});
    }
}

void scheduling(Node *root, std::vector<NodePairVec> &levels, int mode)       // progressive.cpp:109-124
{
    levels.clear();
    std::stack<Node *> post;
    root->collectPostOrder(post);
    std::vector<std::pair<NodePair, int>> pairs;
    getProgressivePairs(pairs, post, root->grpID, mode);
    for (auto &h : pairs) {
        if (levels.size() < (size_t)h.second + 1) levels.resize(h.second + 1);
        levels[h.second].push_back(h.first);
    }
}

// progressive.cpp:126-172: leaves take their sequence; an internal node adopts the result held by its already aligned child
static void materialise(Node *n, Node *partner, SequenceDB *db)
{
    if (n->is_leaf() && n->seqsIncluded.empty()) {
        auto *s = db->name_map.find(n->identifier)->second;      // (every leaf has its sequence: io::readSequences checked)
        n->seqsIncluded.push_back(s->id);
        n->alnLen = s->len;
        n->alnNum = 1;
        n->alnWeight = s->weight;
    } else if (n->seqsIncluded.empty()) {
        const int grp = n->grpID;
        for (Node *c : n->children) {
            if ((c->grpID == -1 || c->grpID == grp) && c->identifier != partner->identifier) {
                n->msaFreq = c->msaFreq;
                c->msaFreq.clear();
                n->cacheId = c->cacheId;
                c->cacheId = -1;
                n->seqsIncluded = std::move(c->seqsIncluded);      // (the reference copies; nothing reads the child's list again: its pair is done)
                c->seqsIncluded.clear();
                n->alnLen = c->alnLen;
                n->alnNum = c->alnNum;
                n->alnWeight = c->alnWeight;
                break;
            }
        }
    }
}

void updateNode(Tree *, NodePairVec &nodes, SequenceDB *database)
{
    // (the pairs of a level touch disjoint nodes, and name_map is only read: the wide leaf levels of a 100 000-leaf tree spent 50 ms here in one thread)
    const int n = (int)nodes.size();
#pragma omp parallel for schedule(static) if (n >= 512)
    for (int i = 0; i < n; ++i) {
        materialise(nodes[i].first, nodes[i].second, database);
        materialise(nodes[i].second, nodes[i].first, database);
    }
}

void progressiveAlignment(Tree *T, SequenceDB *database, Option *option, std::vector<NodePairVec> &levels, Params &param, alnFunction kernel)
{
    int level = 0;
    if (option->printDetail) std::cerr << "Total " << levels.size() << " levels.\n";
    // a sharded run on the device-resident kernel takes the levels below a cut of the tree by subtree ownership (gpu/align_owned.cpp)
    size_t done = 0;
    if (database->ownedPrefix && database->currentTask == 0) done = database->ownedPrefix(T, levels, option, param);
    for (auto m : levels) {                     // serial over levels: tree dependency (progressive.cpp:177)
        if ((size_t)level < done) { ++level; continue; }
        auto t0 = std::chrono::high_resolution_clock::now();
        updateNode(T, m, database);
        auto t1 = std::chrono::high_resolution_clock::now();
        kernel(T, m, database, option, param);
        auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        if (option->printDetail)
            std::cerr << "Level " << level + 1 << ", aligned " << m.size() << (m.size() > 1 ? " pairs in " : " pair in ") << ms << " ms (nodes "
                      << std::chrono::duration<double, std::milli>(t1 - t0).count() << " ms)\n";
        ++level;
    }
}

// progressive.cpp:194-230: expand members that were compressed into a group path
void updateAlignment(Node *node, SequenceDB *database)
{
    const int n = (int)database->sequences.size();
#pragma omp parallel for schedule(dynamic, 16)
    for (int idx = 0; idx < n; ++idx) {
        auto *s = database->sequences[idx];
        if (s->subtreeIdx >= -1) continue;
        const alnPath &aln = database->subtreeAln[s->subtreeIdx];
        s->memCheck((int)aln.size());
        const char *from = s->alnStorage[s->storage];
        char *to = s->alnStorage[1 - s->storage];
        int org = 0;
        for (size_t k = 0; k < aln.size(); ++k) to[k] = (aln[k] == 0) ? from[org++] : '-';
        s->len = (int)aln.size();
        s->changeStorage();
    }
    std::vector<int> members;
    for (int sIdx : node->seqsIncluded)
        if (sIdx >= 0) members.push_back(sIdx);
    for (auto *s : database->sequences)
        if (s->subtreeIdx < 0) members.push_back(s->id);
    node->seqsIncluded = members;
}

// ---- subtree ownership of a sharded run (gpu/align_owned.cpp uses it; pure: unit-tested in tests/host_kats.cpp) ----
// The cut is the highest level of the schedule that still leaves 8 subtrees per rank; the subtrees below it are the classes of the nodes the pairs of
// levels <= cut connect (a pair connects its two operands; an internal operand is connected to the children it adopts its content from,
// progressive.cpp:126-172), dealt longest-first by their number of pairs.  No cut (cut = -1) when the tree is too small for that.
namespace {
struct UnionFind {
    std::vector<int> p;
    int find(int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(int a, int b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};
}
OwnershipPlan planOwnership(Tree *T, const std::vector<NodePairVec> &levels, int world)
{
    OwnershipPlan plan;
    const int nLevels = (int)levels.size();
    if (world <= 1 || nLevels < 2) return plan;
    const long long want = 8ll * world;
    std::vector<long long> above((size_t)nLevels + 1, 0);
    for (int l = nLevels - 1; l >= 0; --l) above[l] = above[l + 1] + (long long)levels[l].size();      // pairs at levels >= l
    int cut = -1;
    for (int l = nLevels - 2; l >= 0; --l)
        if (above[l + 1] + 1 >= want) { cut = l; break; }
    if (cut < 0) return plan;
    std::unordered_map<const Node *, int> idOf;
    auto id = [&](const Node *n) { auto it = idOf.find(n); if (it != idOf.end()) return it->second; const int k = (int)idOf.size(); idOf.emplace(n, k); return k; };
    const int grp = T->root->grpID;
    auto adopts = [&](const Node *c) { return c->grpID == -1 || c->grpID == grp; };
    for (int l = 0; l <= cut; ++l)
        for (auto &pr : levels[l])
            for (const Node *x : {pr.first, pr.second}) {
                id(x);
                if (!x->is_leaf()) for (const Node *c : x->children) if (adopts(c)) id(c);
            }
    UnionFind uf;
    uf.p.resize(idOf.size());
    for (size_t k = 0; k < uf.p.size(); ++k) uf.p[k] = (int)k;
    for (int l = 0; l <= cut; ++l)
        for (auto &pr : levels[l]) {
            uf.unite(idOf[pr.first], idOf[pr.second]);
            for (const Node *x : {pr.first, pr.second})
                if (!x->is_leaf()) for (const Node *c : x->children) if (adopts(c)) uf.unite(idOf[x], idOf[c]);
        }
    // (numbering by first appearance in the schedule: the same on every rank)
    std::unordered_map<int, long long> cost;
    std::vector<int> firstSeen;
    for (int l = 0; l <= cut; ++l)
        for (auto &pr : levels[l]) { const int c = uf.find(idOf[pr.first]); if (!cost.count(c)) firstSeen.push_back(c); cost[c] += 1; }
    std::vector<std::pair<long long, int>> order;      // (pairs, rank of first appearance)
    for (size_t k = 0; k < firstSeen.size(); ++k) order.push_back({cost[firstSeen[k]], (int)k});
    std::sort(order.begin(), order.end(), [](const std::pair<long long, int> &a, const std::pair<long long, int> &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    std::unordered_map<int, int> ownerOf;
    plan.load.assign((size_t)world, 0);
    for (auto &c : order) {
        const int r = (int)(std::min_element(plan.load.begin(), plan.load.end()) - plan.load.begin());
        ownerOf[firstSeen[(size_t)c.second]] = r;
        plan.load[r] += c.first;
    }
    plan.cut = cut;
    plan.subtrees = (int)order.size();
    plan.owner.resize((size_t)cut + 1);
    for (int l = 0; l <= cut; ++l) {
        plan.owner[l].resize(levels[l].size());
        for (size_t i = 0; i < levels[l].size(); ++i) plan.owner[l][i] = ownerOf[uf.find(idOf[levels[l][i].first])];
    }
    return plan;
}

// progressive.cpp:232-299
void msaOnSubtree(Tree *T, SequenceDB *database, Option *option, Params &param, alnFunction kernel, alnFunction deferredKernel)
{
    auto t0 = std::chrono::high_resolution_clock::now();
    std::cerr << "============================\n";
    std::vector<NodePairVec> levels;
    scheduling(T->root, levels, database->currentTask == 0 ? 0 : 1);
    if (option->printDetail) std::cerr << "Levels scheduled in " << std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count() << " ms\n";
    progressiveAlignment(T, database, option, levels, param, kernel);
    const auto tLoop = std::chrono::high_resolution_clock::now();
    if (database->currentTask == 0) {                       // push the result to the root
        Node *last = levels.back()[0].first;
        T->root->seqsIncluded = last->seqsIncluded;
        if (!last->msaFreq.empty()) T->root->msaFreq = last->msaFreq;
        if (last->cacheId >= 0) { T->root->cacheId = last->cacheId; last->cacheId = -1; }
        T->root->alnLen = last->alnLen;
        T->root->alnNum = last->alnNum;
        T->root->alnWeight = last->alnWeight;
        last->seqsIncluded.clear();
        last->msaFreq.clear();
        // device-resident mode: the rows come back now, unless nothing on the host needs them yet (library caller without a deferred pass) or the
        // deferred pass runs on the resident rows as well (then they come back after it, below)
        const bool stayResident = database->fallback_nodes.empty() ? database->lazyRows : database->residentDeferred;
        if (database->afterMainPass && !stayResident) { database->afterMainPass(T); database->afterMainPass = nullptr; }
    }
    if (database->fallback_nodes.empty()) updateAlignment(T->root, database);
    if (option->printDetail) std::cerr << "After the last level: " << std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - tLoop).count() << " ms\n";
    auto secs = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cerr << "Alignment (length: " << T->root->alnLen << ") completed in " << secs << " s\n";
    if (database->fallback_nodes.empty()) return;

    // second pass: deferred profiles/sequences are aligned to the root one per level (progressive.cpp:270-298)
    database->currentTask = 1;
    int deferredSeqs = 0;
    levels.clear();
    for (Node *bad : database->fallback_nodes) deferredSeqs += (int)bad->seqsIncluded.size();
    std::sort(database->fallback_nodes.begin(), database->fallback_nodes.end(), [&](Node *&a, Node *&b) {
        if (a->alnNum == b->alnNum) return a->getAlnLen(database->currentTask) > b->getAlnLen(database->currentTask);
        return a->alnNum > b->alnNum;
    });
    for (Node *bad : database->fallback_nodes) levels.push_back(NodePairVec(1, {T->root, bad}));
    std::cerr << "Realign profiles that have been deferred. Total profiles/sequences: " << database->fallback_nodes.size() << " / " << deferredSeqs << '\n';
    database->fallback_nodes.clear();
    progressiveAlignment(T, database, option, levels, param, deferredKernel);
    if (database->afterMainPass && !database->lazyRows) { database->afterMainPass(T); database->afterMainPass = nullptr; }
    updateAlignment(T->root, database);
    database->currentTask = 0;
}

}  // namespace progressive
}  // namespace msa

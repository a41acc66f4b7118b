// twilight_amd/csrc/host/phylo.cpp -- guide-tree plumbing needed to reproduce the reference's level batches
// (SURVEY.md section 8f-3).  Behavioural mirror of /root/reference/src/{node,tree,phylogeny,partitionInfo}.cpp for the
// DEFAULT_ALN path; every routine cites the lines it follows.  The node map is a std::unordered_map keyed by name
// exactly like the reference, filled in the same order, because reroot() starts its search at the first leaf the
// map iterates to (tree.cpp:601-605).
#include "twl_host.hpp"

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <limits>
#include <queue>

namespace phylogeny {

// node.cpp:7-21
Node::Node(const std::string &id, float len) : identifier(id), parent(nullptr), branchLength(len), level(1) {}
Node::Node(const std::string &id, Node *par, float len) : identifier(id), parent(par), branchLength(len), level(par->level + 1)
{
    par->children.push_back(this);
}

// node.cpp:58-70: stack order such that popping yields children before parents; only same-group children are followed
void Node::collectPostOrder(std::stack<Node *> &postStack)
{
    std::stack<Node *> work;
    work.push(this);
    while (!work.empty()) {
        Node *cur = work.top();
        work.pop();
        postStack.push(cur);
        for (int i = (int)cur->children.size() - 1; i >= 0; --i)
            if (cur->children[i]->grpID == cur->grpID) work.push(cur->children[i]);
    }
}

void updateLevels(Node *node, size_t currentLevel)      // tree.cpp:704-710
{
    if (!node) return;
    node->level = currentLevel;
    for (Node *c : node->children) updateLevels(c, currentLevel + 1);
}

static void setGroup(Node *node, int from, int to)      // partitionInfo.cpp:45-53
{
    if (node->grpID != from) return;
    node->grpID = to;
    for (Node *c : node->children) setGroup(c, from, to);
}
void assignSinglePartition(Node *root) { setGroup(root, root->grpID, 0); }

// ---- Newick (tree.cpp:15-223) -------------------------------------------------------------------------------
// The reference cuts the string at commas (outside quotes) and scans each piece once.  What is observable:
//  * internal nodes are named node_1, node_2, ... in order of their '(';
//  * map insertion order: for every piece, its new internal nodes, then its leaf;
//  * a branch length is the number after ':'; an element followed by ')' with length 0/missing gets 1.0, an element
//    followed by ',' or the end keeps 0 (fixed up afterwards); a ')' with no ':' since the previous one reuses the
//    previous number of the same piece;
//  * afterwards zero lengths become the smallest positive length (or 1.0 if all are zero); the root gets 0.
static void splitOutsideQuotes(const std::string &s, std::vector<std::string> &pieces)
{
    size_t start = 0, pending = std::string::npos;
    for (size_t pos = 0; (pos = s.find(',', start)) != std::string::npos; start = pos + 1) {
        if (pending == std::string::npos) {
            std::string sub = s.substr(start, pos - start);
            if (std::count(sub.begin(), sub.end(), '\'') % 2 == 1) pending = start;
            else pieces.emplace_back(std::move(sub));
        } else {
            std::string sub = s.substr(pending, pos - pending);
            if (std::count(sub.begin(), sub.end(), '\'') % 2 == 0) { pending = std::string::npos; pieces.emplace_back(std::move(sub)); }
        }
    }
    std::string last = s.substr(start);
    if (!last.empty()) pieces.push_back(std::move(last));
}

void Tree::parseNewick(std::string &newick)
{
    while (!newick.empty() && newick.back() == ' ') newick.pop_back();
    size_t lead = newick.find_first_not_of(' ');
    if (lead != std::string::npos && lead > 0) newick = newick.substr(lead);

    std::vector<std::string> pieces;
    splitOutsideQuotes(newick, pieces);

    struct Piece { std::string leaf; size_t opens = 0, closes = 0; };
    std::vector<Piece> parsed;
    parsed.reserve(pieces.size());
    std::vector<std::queue<float>> lenAtLevel(128);
    size_t level = 0;

    for (const std::string &piece : pieces) {
        Piece pc;
        size_t leafDepth = 0;
        bool afterName = false, inNumber = false, inQuote = false, quoted = false;
        std::string number;
        for (char c : piece) {
            if (inQuote) { pc.leaf += c; if (c == '\'') inQuote = false; }
            else if (c == '\'') { inQuote = true; quoted = true; pc.leaf += c; }
            else if (c == ':') { afterName = true; number.clear(); inNumber = true; }
            else if (c == '(') { pc.opens++; level++; if (lenAtLevel.size() <= level) lenAtLevel.resize(level * 2); }
            else if (c == ')') {
                afterName = true;
                pc.closes++;
                float len = number.empty() ? 0.0f : std::stof(number);
                if (len == 0) len = 1.0f;
                lenAtLevel[level].push(len);
                level--;
                inNumber = false;
            }
            else if (!afterName) { pc.leaf += c; inNumber = false; leafDepth = level; }
            else if (inNumber) { if (isdigit((unsigned char)c) || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') number += c; }
        }
        if (quoted && pc.leaf.size() >= 2 && pc.leaf.front() == '\'' && pc.leaf.back() == '\'') pc.leaf = pc.leaf.substr(1, pc.leaf.size() - 2);
        lenAtLevel[level].push(number.empty() ? 0.0f : std::stof(number));
        m_maxDepth = std::max(m_maxDepth, leafDepth);
        m_meanDepth += leafDepth;
        parsed.push_back(std::move(pc));
    }
    m_meanDepth /= parsed.size();
    if (level != 0) { fprintf(stderr, "ERROR: incorrect Newick format!\n"); exit(1); }
    m_numLeaves = parsed.size();

    std::stack<Node *> open;
    Node *top = nullptr;
    for (Piece &pc : parsed) {
        for (size_t j = 0; j < pc.opens; ++j) {
            std::string nid = newInternalNodeId();
            Node *n = open.empty() ? new Node(nid, lenAtLevel[level].front()) : new Node(nid, open.top(), lenAtLevel[level].front());
            if (open.empty()) top = n;
            lenAtLevel[level].pop();
            level++;
            n->grpID = -1;
            allNodes[nid] = n;
            open.push(n);
        }
        if (allNodes.find(pc.leaf) != allNodes.end()) {
            printf("WARNING: duplicate leaf names found in the tree! Leaf name: %s. All duplicate leaves will be removed during processing.\n", pc.leaf.c_str());
            pc.leaf += "_dup_" + std::to_string(allNodes.size());
        }
        Node *leafNode = new Node(pc.leaf, open.top(), lenAtLevel[level].front());
        leafNode->grpID = -1;
        allNodes[pc.leaf] = leafNode;
        lenAtLevel[level].pop();
        for (size_t j = 0; j < pc.closes; ++j) { open.pop(); level--; }
    }
    if (!top) { fprintf(stderr, "WARNING: Tree found empty!\n"); exit(1); }
    top->branchLength = 0;
    root = top;

    float minLen = std::numeric_limits<float>::max();
    bool allZero = true;
    for (auto &kv : allNodes) {
        float b = kv.second->branchLength;
        if (b > 0 && b < minLen) minLen = b;
        if (b > 0) allZero = false;
    }
    for (auto &kv : allNodes) {
        Node *n = kv.second;
        if (n->identifier == root->identifier) continue;
        if (allZero) n->branchLength = 1.0f;
        else if (n->branchLength == 0) n->branchLength = minLen;
    }
    calLeafNum();
    calSeqWeight();
}

Tree::Tree(const std::string &file)        // tree.cpp:225-237 (first line of the file only)
{
    std::ifstream in(file);
    if (!in) { fprintf(stderr, "Error: Failed to open file: %s\n", file.c_str()); exit(1); }
    std::string newick;
    std::getline(in, newick);
    parseNewick(newick);
}

// tree.cpp:239-272: copy the nodes of node's group in stack (pre-)order into a fresh map, then reroot
Tree::Tree(Node *node, bool doReroot)
{
    Node *r = new Node(node->identifier, node->branchLength);
    const int grp = node->grpID;
    r->grpID = -1;
    allNodes[node->identifier] = r;
    root = r;
    std::stack<Node *> work;
    work.push(node);
    while (!work.empty()) {
        Node *cur = work.top();
        if (cur->identifier != root->identifier) {
            Node *cp = new Node(cur->identifier, allNodes[cur->parent->identifier], cur->branchLength);
            cp->grpID = -1;
            cp->level = cur->level - (node->level - 1);
            cp->weight = cur->weight;
            allNodes[cur->identifier] = cp;
        }
        work.pop();
        for (int i = (int)cur->children.size() - 1; i >= 0; --i)
            if (cur->children[i]->grpID == grp) work.push(cur->children[i]);
    }
    int maxInternal = 0;
    for (auto &kv : allNodes)
        if (!kv.second->is_leaf()) maxInternal = std::max(std::stoi(kv.first.substr(5)), maxInternal);
    m_currInternalNode = maxInternal;
    if (doReroot) reroot();
    else { calLeafNum(); calSeqWeight(); }
}

Tree::~Tree()
{
    for (auto &kv : allNodes) delete kv.second;
}

void Tree::calLeafNum()                    // tree.cpp:295-315
{
    std::stack<Node *> post;
    root->collectPostOrder(post);
    while (!post.empty()) {
        Node *cur = post.top();
        post.pop();
        if (cur->is_leaf()) allNodes[cur->identifier]->numLeaves = 1;
        else {
            int leaves = 0;
            for (Node *c : cur->children) leaves += (int)c->numLeaves;
            allNodes[cur->identifier]->numLeaves = leaves;
        }
    }
    m_numLeaves = root->numLeaves;
}

void Tree::calSeqWeight()                  // tree.cpp:317-343: sum of branch/leaves-below up to the root, scaled to max 1
{
    float maxW = 0;
    for (auto &kv : allNodes) {
        if (!kv.second->is_leaf()) continue;
        float w = 0;
        for (Node *cur = kv.second; cur != nullptr; cur = cur->parent) w += cur->branchLength / cur->numLeaves;
        allNodes[kv.second->identifier]->weight = w;
        if (w > maxW) maxW = w;
    }
    const float norm = maxW / 1.0;
    for (auto &kv : allNodes)
        if (kv.second->is_leaf()) allNodes[kv.second->identifier]->weight /= norm;
}

void Tree::convert2binaryTree()            // tree.cpp:528-586
{
    std::stack<Node *> post;
    root->collectPostOrder(post);
    while (!post.empty()) {
        Node *node = post.top();
        if (node->children.size() > 2) {
            const int grp = node->grpID;
            std::vector<Node *> cur = node->children;
            while (cur.size() > 2) {
                std::vector<Node *> next;
                for (size_t i = 0; i + 1 < cur.size(); i += 2) {
                    std::string name = newInternalNodeId();
                    Node *joint = new Node(name, 0.0f);
                    joint->children.push_back(cur[i]);
                    joint->children.push_back(cur[i + 1]);
                    joint->grpID = grp;
                    allNodes[name] = joint;
                    cur[i]->parent = joint;
                    cur[i + 1]->parent = joint;
                    next.push_back(joint);
                }
                if (cur.size() % 2 == 1) next.push_back(cur.back());
                cur = next;
            }
            assert(cur.size() == 2);
            node->children = {cur[0], cur[1]};
            cur[0]->parent = node;
            cur[1]->parent = node;
        } else if (node->children.size() == 1 && node->parent != nullptr) {
            for (size_t c = 0; c < node->parent->children.size(); ++c)
                if (node->parent->children[c]->identifier == node->identifier) {
                    node->parent->children[c] = node->children[0];
                    node->children[0]->branchLength += node->branchLength;
                    node->children[0]->parent = node->parent;
                    break;
                }
        } else if (node->children.empty() && !node->is_leaf() && !node->seqsIncluded.empty()) {
            std::vector<Node *> keep;
            for (Node *c : node->parent->children)
                if (c->identifier != node->identifier) keep.push_back(c);
            node->parent->children = keep;
        }
        post.pop();
    }
    updateLevels(root, 1);
}

// tree.cpp:588-696: move the root to the middle of the tree's diameter (in edges) so that sibling levels are wide
void Tree::reroot()
{
    int depth0 = 0, depth1 = 0, depth2 = 0;
    for (auto &kv : allNodes) depth0 = std::max<int>(depth0, (int)kv.second->level);
    convert2binaryTree();
    for (auto &kv : allNodes) depth1 = std::max<int>(depth1, (int)kv.second->level);
    Node *start = nullptr;
    for (auto &kv : allNodes)
        if (kv.second->is_leaf()) { start = kv.second; break; }

    auto farthestFrom = [&](Node *from, std::unordered_map<Node *, Node *> &cameFrom) {
        std::queue<Node *> q;
        std::unordered_map<Node *, int> dist;
        cameFrom.clear();
        q.push(from);
        dist[from] = 0;
        cameFrom[from] = nullptr;
        Node *far = from;
        while (!q.empty()) {
            Node *u = q.front();
            q.pop();
            std::vector<Node *> neigh = u->children;
            if (u->parent) neigh.push_back(u->parent);
            for (Node *v : neigh) {
                if (dist.count(v)) continue;
                dist[v] = dist[u] + 1;
                cameFrom[v] = u;
                q.push(v);
                if (dist[v] > dist[far]) far = v;
            }
        }
        return far;
    };
    std::unordered_map<Node *, Node *> fromA, fromB;
    Node *A = farthestFrom(start, fromA);
    Node *B = farthestFrom(A, fromB);
    std::vector<Node *> path;
    for (Node *cur = B; cur != nullptr; cur = fromB[cur]) path.push_back(cur);
    std::reverse(path.begin(), path.end());
    Node *newRoot = path[path.size() / 2];
    if (newRoot->identifier == root->identifier) return;      // note: leaves numLeaves/weights/m_numLeaves untouched (:658)

    // Turn the path new root -> old root upside down, walking it from the bottom: every edge keeps its length and its two ends trade places,
    // so a node of the path hangs under the neighbour we came from, carries the length of the edge between them, loses that neighbour as a
    // child and gains its former parent as its last child (tree.cpp:657-674 builds the same links from the top; child order is observable:
    // convert2binaryTree and the level schedule walk children in order).
    {
        Node *came = nullptr;
        float edge = 0.0f;                                      // length of the edge between `came` and the node in hand
        for (Node *cur = newRoot; cur != nullptr;) {
            Node *up = cur->parent;
            const float upEdge = cur->branchLength;
            if (came) cur->children.erase(std::remove(cur->children.begin(), cur->children.end(), came), cur->children.end());
            if (up) cur->children.push_back(up);
            cur->parent = came;
            cur->branchLength = came ? edge : 0.0f;
            came = cur; edge = upEdge; cur = up;
        }
    }
    updateLevels(newRoot, 1);
    // The root keeps its NAME (tree.cpp:679-686): the node that is the root now and the one that was trade identifiers; both keys exist, so
    // the two map slots trade their nodes.
    std::swap(root->identifier, newRoot->identifier);
    std::swap(allNodes.at(root->identifier), allNodes.at(newRoot->identifier));
    root = newRoot;
    convert2binaryTree();
    calLeafNum();
    calSeqWeight();
    for (auto &kv : allNodes) depth2 = std::max<int>(depth2, (int)kv.second->level);
    std::cerr << "======== Tree Depth ========\nOriginal: " << depth0 << "\nBinary: " << depth1 << "\nReroot: " << depth2 << '\n';
}

// tree.cpp:378-494: keep only the leaves in `seqs`, contracting unary chains
Tree *Tree::prune(std::unordered_set<std::string> &seqs)
{
    Tree *pT = new Tree();
    pT->root = new Node(root->identifier, root->branchLength);
    pT->root->grpID = -1;
    pT->allNodes[pT->root->identifier] = pT->root;

    std::unordered_map<std::string, bool> keep;
    for (auto &kv : allNodes)
        if (kv.second->is_leaf()) keep[kv.second->identifier] = seqs.count(kv.second->identifier) > 0;
    std::function<bool(Node *)> mark = [&](Node *n) -> bool {
        if (n->is_leaf()) return keep[n->identifier];
        bool any = false;
        for (Node *c : n->children)
            if (mark(c)) any = true;
        keep[n->identifier] = any;
        return any;
    };
    mark(root);

    std::function<void(Node *, Node *)> build = [&](Node *orig, Node *newParent) {
        if (!keep[orig->identifier]) return;
        if (orig->identifier == root->identifier) {
            for (Node *c : root->children) build(c, root);
            return;
        }
        std::vector<Node *> kids;
        for (Node *c : allNodes[orig->identifier]->children)
            if (keep[c->identifier]) kids.push_back(c);
        if (kids.empty()) {
            if (!orig->is_leaf()) return;
            Node *n = new Node(orig->identifier, pT->allNodes[newParent->identifier], orig->branchLength);
            n->grpID = -1;
            pT->allNodes[n->identifier] = n;
        } else if (kids.size() == 1) {
            Node *only = kids[0];
            float combined = orig->branchLength;
            while (true) {
                std::vector<Node *> below;
                combined += only->branchLength;
                for (Node *c : only->children)
                    if (keep[c->identifier]) below.push_back(c);
                if (below.size() > 1 || (below.empty() && only->is_leaf())) {
                    Node *n = new Node(only->identifier, pT->allNodes[newParent->identifier], combined);
                    n->grpID = -1;
                    pT->allNodes[n->identifier] = n;
                    break;
                }
                if (below.empty()) return;
                only = below[0];
            }
            for (Node *g : allNodes[only->identifier]->children) build(g, allNodes[only->identifier]);
        } else {
            Node *n = new Node(orig->identifier, pT->allNodes[newParent->identifier], orig->branchLength);
            n->grpID = -1;
            pT->allNodes[n->identifier] = n;
            for (Node *c : allNodes[orig->identifier]->children) build(c, allNodes[orig->identifier]);
        }
    };
    build(pT->root, nullptr);

    pT->calLeafNum();
    pT->calSeqWeight();
    std::cerr << "Number of Leaves: " << m_numLeaves << " (before pruning) -> " << pT->m_numLeaves << " (after pruning)\n";
    if (pT->m_numLeaves == 0) { std::cerr << "ERROR: No sequences from the input sequence file are found in the tree file.\n"; exit(1); }
    if (pT->m_numLeaves != seqs.size())
        std::cerr << "WARNING: " << (seqs.size() - pT->m_numLeaves) << " sequences are missing from the tree and will be ignored.\n";
    return pT;
}

void pruneTree(Tree *&T, std::unordered_set<std::string> &seqs)      // phylogeny.cpp:5-10
{
    Tree *p = T->prune(seqs);
    delete T;
    T = p;
}

}  // namespace phylogeny

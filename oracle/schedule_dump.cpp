// oracle/schedule_dump.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Runs the host mirror up to the level schedule (tree -> partition -> reroot -> read sequences -> scheduling; the part that the two
// reference pins validate through the recorded pairs-per-level) and dumps what an independent implementation of everything AFTER it
// needs: the sequences with their weights and quality flags, the subtree's nodes, and the level batches.  oracle/msa_replay.py
// (numpy + the oracle DP) replays the progressive alignment from this dump without sharing any code with helpers.cpp /
// progressive.cpp / align_*.cpp, so that the host mirror's rows are checked against something outside itself.
//   schedule_dump <CLI flags as for twilight-mi355x> > dump.json
#include "../twilight_amd/csrc/host/twl_host.hpp"

#include <cstdio>
#include <iostream>
#include <stack>

static void jstr(const std::string &s)
{
    putchar('"');
    for (char c : s) { if (c == '"' || c == '\\') putchar('\\'); putchar(c); }
    putchar('"');
}

int main(int argc, char **argv)
{
    msa::Option option;
    if (!msa::parseCommandLine(argc, argv, option)) return 1;
    msa::SequenceDB db;
    db.updateSeqTh = option.updateSeqTh;
    msa::Tree *T = new msa::Tree(option.treeFile);
    phylogeny::assignSinglePartition(T->root);
    msa::Tree *subT = new msa::Tree(T->root, option.reroot);
    msa::io::readSequences(option.seqFile, &db, &option, subT);
    std::vector<msa::NodePairVec> levels;
    msa::progressive::scheduling(subT->root, levels, 0);
    printf("{\"type\":\"%c\",\"root\":", option.type);
    jstr(subT->root->identifier);
    printf(",\n\"sequences\":[");
    for (size_t i = 0; i < db.sequences.size(); ++i) {
        auto *s = db.sequences[i];
        printf("%s\n{\"id\":%d,\"name\":", i ? "," : "", s->id);
        jstr(s->name);
        printf(",\"weight\":%.9g,\"low_quality\":%d,\"subtree_idx\":%d,\"seq\":", s->weight, s->lowQuality ? 1 : 0, s->subtreeIdx);
        jstr(std::string(s->alnStorage[s->storage], (size_t)s->len));
        printf("}");
    }
    printf("],\n\"nodes\":{");
    bool first = true;
    std::stack<msa::Node *> st;
    st.push(subT->root);
    while (!st.empty()) {
        msa::Node *n = st.top(); st.pop();
        printf("%s\n", first ? "" : ",");
        first = false;
        jstr(n->identifier);
        printf(":{\"leaf\":%d,\"grp\":%d,\"children\":[", n->is_leaf() ? 1 : 0, n->grpID);
        for (size_t c = 0; c < n->children.size(); ++c) { if (c) putchar(','); jstr(n->children[c]->identifier); st.push(n->children[c]); }
        printf("]}");
    }
    printf("},\n\"levels\":[");
    for (size_t l = 0; l < levels.size(); ++l) {
        printf("%s\n[", l ? "," : "");
        for (size_t i = 0; i < levels[l].size(); ++i) { if (i) putchar(','); putchar('['); jstr(levels[l][i].first->identifier); putchar(','); jstr(levels[l][i].second->identifier); putchar(']'); }
        printf("]");
    }
    printf("]}\n");
    return 0;
}

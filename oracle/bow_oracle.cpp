/* ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the DBoW2 vocabulary transform the reference runs per frame
 * (Frame::ComputeBoW, src/Frame.cc:348-355 -> ORBVocabulary::transform(features, BowVector, FeatureVector, 4)), for
 * tests/ and bench.py's cpu_baseline.  Never linked into or called by the product library.
 * PARITY UNPINNED: the reference ships no tests, golden vectors or vocabulary file (ORBvoc.txt is external).
 *
 * What it follows (reference tree Thirdparty/DBoW2/DBoW2):
 *   TemplatedVocabulary.h:1338-1425  loadFromTextFile (incl. its `while(!f.eof())` extra node after a trailing newline)
 *   TemplatedVocabulary.h:1127-1194  transform(features, BowVector&, FeatureVector&, levelsup)
 *   TemplatedVocabulary.h:1218-1259  transform(feature, word_id, weight, nid, levelsup): the tree descent
 *   FORB.cpp:81-101                  FORB::distance, :122-137 FORB::fromString
 *   BowVector.cpp:32-89              addWeight / addIfNotExist / normalize
 *   FeatureVector.cpp:30-45          addFeature
 *   ScoringObject.h:74-91            which scoring types normalise, and with which norm
 * std::map keeps the reference's iteration (and therefore summation) order. */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

enum { TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 };
enum { L1_NORM = 0, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };

struct Node {
    unsigned id = 0;
    double weight = 0;
    std::vector<unsigned> children;
    unsigned parent = 0;
    uint8_t descriptor[32] = {0};
    unsigned word_id = 0;
    bool isLeaf() const { return children.empty(); }
};

struct Vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<Node> nodes;
    std::vector<unsigned> words; /* node id of every word */
    bool empty() const { return words.empty(); }
};

int forb_distance(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        unsigned int v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void add_node(Vocabulary& v, int pid, int nIsLeaf, const uint8_t* desc, double weight)
{
    unsigned nid = (unsigned)v.nodes.size();
    v.nodes.resize(v.nodes.size() + 1);
    v.nodes[nid].id = nid;
    v.nodes[nid].parent = (unsigned)pid;
    v.nodes[pid].children.push_back(nid);
    std::memcpy(v.nodes[nid].descriptor, desc, 32);
    v.nodes[nid].weight = weight;
    if (nIsLeaf > 0) {
        unsigned wid = (unsigned)v.words.size();
        v.words.push_back(nid);
        v.nodes[nid].word_id = wid;
    }
}

/* transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1218-1259.  *nid is left alone when the level is
 * never reached (a leaf above level L - levelsup): the caller's variable is uninitialised there in the reference; this
 * build defines it as the leaf reached. */
void transform_one(const Vocabulary& v, const uint8_t* feature, unsigned& word_id, double& weight, unsigned* nid, int levelsup)
{
    const int nid_level = v.L - levelsup;
    bool nid_set = false;
    if (nid_level <= 0 && nid != nullptr) { *nid = 0; nid_set = true; }
    unsigned final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const std::vector<unsigned>& nodes = v.nodes[final_id].children;
        final_id = nodes[0];
        double best_d = forb_distance(feature, v.nodes[final_id].descriptor);
        for (size_t i = 1; i < nodes.size(); i++) {
            unsigned id = nodes[i];
            double d = forb_distance(feature, v.nodes[id].descriptor);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (nid != nullptr && current_level == nid_level) { *nid = final_id; nid_set = true; }
    } while (!v.nodes[final_id].isLeaf());
    if (nid != nullptr && !nid_set) *nid = final_id;
    word_id = v.nodes[final_id].word_id;
    weight = v.nodes[final_id].weight;
}

bool must_normalize(int scoring, int& norm /*1 = L1, 2 = L2*/)
{
    norm = scoring == L2_NORM ? 2 : 1;
    return scoring != DOT_PRODUCT;
}

} // namespace

extern "C" {

void* oracle_voc_create(int k, int L, int scoring, int weighting)
{
    Vocabulary* v = new Vocabulary();
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->nodes.resize(1);
    return v;
}

void oracle_voc_add_node(void* h, int parent, int is_leaf, const uint8_t* desc, double weight)
{
    add_node(*(Vocabulary*)h, parent, is_leaf, desc, weight);
}

void* oracle_voc_load_text(const char* filename)
{
    std::ifstream f;
    f.open(filename);
    if (!f.is_open() || f.eof()) return nullptr;
    Vocabulary* v = new Vocabulary();
    std::string s;
    std::getline(f, s);
    std::stringstream ss;
    ss << s;
    int n1 = 0, n2 = 0;
    ss >> v->k;
    ss >> v->L;
    ss >> n1;
    ss >> n2;
    if (v->k < 0 || v->k > 20 || v->L < 1 || v->L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        delete v;
        return nullptr;
    }
    v->scoring = n1; v->weighting = n2;
    v->nodes.resize(1);
    while (!f.eof()) {
        std::string snode;
        std::getline(f, snode);
        std::stringstream ssnode;
        ssnode << snode;
        int pid = 0;
        ssnode >> pid;
        int nIsLeaf = 0;
        ssnode >> nIsLeaf;
        std::stringstream ssd;
        for (int iD = 0; iD < 32; iD++) {
            std::string sElement;
            ssnode >> sElement;
            ssd << sElement << " ";
        }
        uint8_t desc[32] = {0}; /* FORB::fromString on a fresh zero... a.create() leaves bytes unset on failure; zeros here */
        {
            std::stringstream sd(ssd.str());
            for (int i = 0; i < 32; ++i) {
                int n = 0;
                sd >> n;
                if (!sd.fail()) desc[i] = (unsigned char)n;
            }
        }
        double weight = 0;
        ssnode >> weight;
        if (pid < 0 || pid >= (int)v->nodes.size()) { delete v; return nullptr; }
        add_node(*v, pid, nIsLeaf, desc, weight);
    }
    return v;
}

void oracle_voc_destroy(void* h) { delete (Vocabulary*)h; }

void oracle_voc_info(void* h, int32_t* out)
{
    Vocabulary* v = (Vocabulary*)h;
    out[0] = v->k; out[1] = v->L; out[2] = v->scoring; out[3] = v->weighting;
    out[4] = (int32_t)v->nodes.size(); out[5] = (int32_t)v->words.size();
}

/* per-feature descent */
void oracle_voc_transform_features(void* h, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                                   double* weight)
{
    Vocabulary* v = (Vocabulary*)h;
    for (int i = 0; i < n; i++) {
        unsigned id = 0, nid = 0;
        double w = 0;
        if (!v->empty()) transform_one(*v, desc + 32 * (size_t)i, id, w, &nid, levelsup);
        word_id[i] = (int32_t)id; node_id[i] = (int32_t)nid; weight[i] = w;
    }
}

/* transform(features, BowVector, FeatureVector, levelsup), flattened in map order.  fv_offset has *nfv + 1 entries. */
void oracle_voc_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* bow_word, double* bow_value, int32_t* nbow,
                          uint32_t* fv_node, int32_t* fv_offset, uint32_t* fv_feature, int32_t* nfv)
{
    Vocabulary* v = (Vocabulary*)h;
    std::map<unsigned, double> bow;
    std::map<unsigned, std::vector<unsigned>> fv;
    *nbow = 0; *nfv = 0; fv_offset[0] = 0;
    if (v->empty()) return;
    int norm;
    bool must = must_normalize(v->scoring, norm);
    if (v->weighting == TF || v->weighting == TF_IDF) {
        for (int i = 0; i < n; i++) {
            unsigned id, nid;
            double w;
            transform_one(*v, desc + 32 * (size_t)i, id, w, &nid, levelsup);
            if (w > 0) {
                auto it = bow.lower_bound(id);
                if (it != bow.end() && !(id < it->first)) it->second += w;
                else bow.insert(it, std::make_pair(id, w));
                fv[nid].push_back((unsigned)i);
            }
        }
        if (!bow.empty() && !must) {
            const double nd = (double)bow.size();
            for (auto& e : bow) e.second /= nd;
        }
    } else {
        for (int i = 0; i < n; i++) {
            unsigned id, nid;
            double w;
            transform_one(*v, desc + 32 * (size_t)i, id, w, &nid, levelsup);
            if (w > 0) {
                auto it = bow.lower_bound(id);
                if (it == bow.end() || (id < it->first)) bow.insert(it, std::make_pair(id, w));
                fv[nid].push_back((unsigned)i);
            }
        }
    }
    if (must) {
        double nrm = 0.0;
        if (norm == 1) {
            for (auto& e : bow) nrm += std::fabs(e.second);
        } else {
            for (auto& e : bow) nrm += e.second * e.second;
            nrm = std::sqrt(nrm);
        }
        if (nrm > 0.0)
            for (auto& e : bow) e.second /= nrm;
    }
    int k = 0;
    for (auto& e : bow) { bow_word[k] = e.first; bow_value[k] = e.second; k++; }
    *nbow = k;
    k = 0;
    int off = 0;
    for (auto& e : fv) {
        fv_node[k] = e.first;
        fv_offset[k] = off;
        for (unsigned f : e.second) fv_feature[off++] = f;
        k++;
    }
    fv_offset[k] = off;
    *nfv = k;
}

} /* extern "C" */

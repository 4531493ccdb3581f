/* TEST INFRASTRUCTURE ONLY -- C glue around the REFERENCE's own DBoW2::BowVector and DBoW2::FeatureVector classes.
 *
 * The two translation units Thirdparty/DBoW2/DBoW2/BowVector.cpp and FeatureVector.cpp need nothing but the STL, so they are
 * compiled here AS THEY LIE under /root/reference (oracle/Makefile, target `ref`; output only into oracle/_ref/, never copied
 * into the repository).  This file contains no reference code: it includes the reference headers at build time and exposes
 *   - BowVector::addWeight / addIfNotExist over a (word, value) sequence followed by BowVector::normalize(L1 | L2)
 *   - FeatureVector::addFeature over a (node, feature) sequence
 * as flat arrays, so tests/test_oracle_cpu.py can pin oracle/bow_oracle.cpp's restatement of exactly these members (the map's
 * iteration = summation order) against the real classes.  The rest of the vocabulary (TemplatedVocabulary.h, ScoringObject.cpp,
 * FORB.cpp) includes OpenCV headers and is unbuildable in this image. */
#include <cstdint>

#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

/* mode 0: addWeight (TF / TF_IDF accumulation, TemplatedVocabulary.h:1161,1185), 1: addIfNotExist (IDF / BINARY, :1163,1187).
 * norm: 0 none, 1 L1, 2 L2 (ScoringObject.h:74-91 decides which).  Returns the number of entries written (ascending word id). */
int dbow2_ref_bowvector(const uint32_t* word, const double* value, int n, int mode, int norm, uint32_t* out_word, double* out_value)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++) {
        if (mode == 0) v.addWeight(word[i], value[i]);
        else v.addIfNotExist(word[i], value[i]);
    }
    if (norm == 1) v.normalize(DBoW2::L1);
    else if (norm == 2) v.normalize(DBoW2::L2);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) {
        out_word[k] = it->first;
        out_value[k] = it->second;
    }
    return k;
}

/* Returns the number of nodes; offsets[nnodes + 1] index `features` (insertion order inside a node). */
int dbow2_ref_featurevector(const uint32_t* node, const uint32_t* feature, int n, uint32_t* out_node, int32_t* offsets, uint32_t* features)
{
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++) fv.addFeature(node[i], feature[i]);
    int k = 0, o = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        out_node[k] = it->first;
        offsets[k] = o;
        for (size_t j = 0; j < it->second.size(); j++) features[o++] = it->second[j];
    }
    offsets[k] = o;
    return k;
}

} /* extern "C" */

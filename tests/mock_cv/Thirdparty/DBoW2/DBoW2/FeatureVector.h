// TEST INFRASTRUCTURE ONLY -- the two DBoW2 typedef-level facts the ORBmatcher shim relies on (see opencv2/core/core.hpp here)
#ifndef MOCK_DBOW2_FEATUREVECTOR_H
#define MOCK_DBOW2_FEATUREVECTOR_H
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {};
}
#endif

// TEST INFRASTRUCTURE ONLY.
// Stand-in for <boost/graph/graph_selectors.hpp> (Boost is absent from this container): the tag
// types src/CompactUndirectedGraph.hpp:317-319 names.  oracle/_ref only.
#pragma once
namespace boost {
    struct undirectedS {};
    struct directedS {};
    struct bidirectionalS {};
}

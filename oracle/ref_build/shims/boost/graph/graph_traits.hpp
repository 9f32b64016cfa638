// TEST INFRASTRUCTURE ONLY.
// Stand-in for <boost/graph/graph_traits.hpp>: graph_traits<G> forwards to G's nested types, as the
// real primary template does; the category tags src/CompactUndirectedGraph.hpp:317-319 names.
#pragma once
namespace boost {
    struct allow_parallel_edge_tag {};
    struct disallow_parallel_edge_tag {};
    struct adjacency_graph_tag {};
    template<class G> struct graph_traits {
        using vertex_descriptor = typename G::vertex_descriptor;
        using edge_descriptor = typename G::edge_descriptor;
        using vertex_iterator = typename G::vertex_iterator;
        using edge_iterator = typename G::edge_iterator;
        using out_edge_iterator = typename G::out_edge_iterator;
    };
}

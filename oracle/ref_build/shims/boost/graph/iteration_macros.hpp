// TEST INFRASTRUCTURE ONLY.
// Stand-in for <boost/graph/iteration_macros.hpp>: the four loops the reference's method-0 aligner
// uses (src/AlignmentGraph.cpp:274,406,427,446, src/shortestPath.hpp:83,138).  Like the real macros
// they walk [first, last) of vertices(g) / edges(g) / out_edges(v, g) in iterator order with the
// loop variable bound to the dereferenced iterator, and `continue` moves to the next element.
#pragma once
#include <utility>

#define BGL_SHIM_CAT2(a, b) a##b
#define BGL_SHIM_CAT(a, b) BGL_SHIM_CAT2(a, b)
#define BGL_SHIM_FORALL(RANGE, DESCRIPTOR, NAME) \
    for(auto BGL_SHIM_CAT(bglRange, __LINE__) = (RANGE); \
        BGL_SHIM_CAT(bglRange, __LINE__).first != BGL_SHIM_CAT(bglRange, __LINE__).second; \
        ++BGL_SHIM_CAT(bglRange, __LINE__).first) \
        if(bool BGL_SHIM_CAT(bglDone, __LINE__) = false) {} else \
            for(DESCRIPTOR NAME = *BGL_SHIM_CAT(bglRange, __LINE__).first; !BGL_SHIM_CAT(bglDone, __LINE__); BGL_SHIM_CAT(bglDone, __LINE__) = true)

#define BGL_FORALL_VERTICES_T(VNAME, GNAME, GraphType) BGL_SHIM_FORALL(vertices(GNAME), typename GraphType::vertex_descriptor, VNAME)
#define BGL_FORALL_VERTICES(VNAME, GNAME, GraphType) BGL_SHIM_FORALL(vertices(GNAME), GraphType::vertex_descriptor, VNAME)
#define BGL_FORALL_EDGES_T(ENAME, GNAME, GraphType) BGL_SHIM_FORALL(edges(GNAME), typename GraphType::edge_descriptor, ENAME)
#define BGL_FORALL_EDGES(ENAME, GNAME, GraphType) BGL_SHIM_FORALL(edges(GNAME), GraphType::edge_descriptor, ENAME)
#define BGL_FORALL_OUTEDGES_T(UNAME, ENAME, GNAME, GraphType) BGL_SHIM_FORALL(out_edges(UNAME, GNAME), typename GraphType::edge_descriptor, ENAME)
#define BGL_FORALL_OUTEDGES(UNAME, ENAME, GNAME, GraphType) BGL_SHIM_FORALL(out_edges(UNAME, GNAME), GraphType::edge_descriptor, ENAME)

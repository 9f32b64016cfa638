// TEST INFRASTRUCTURE ONLY.
// Stand-in for <boost/pending/disjoint_sets.hpp> (Boost is not installed here),
// restating the published algorithm of boost::disjoint_sets with its default
// find_with_full_path_compression: union by rank (link_sets) and full path
// compression, as used at /root/reference/src/Align4.cpp:814-858.
#ifndef SHIM_BOOST_DISJOINT_SETS_HPP
#define SHIM_BOOST_DISJOINT_SETS_HPP
namespace boost {
template<class RankPA, class ParentPA>
class disjoint_sets {
public:
    disjoint_sets(RankPA r, ParentPA p) : rank(r), parent(p) {}
    template<class E> void make_set(E x) { parent[x] = x; rank[x] = 0; }
    template<class E> E find_set(E x)
    {
        E root = x;
        while(parent[root] != root) root = E(parent[root]);
        while(parent[x] != root) { const E next = E(parent[x]); parent[x] = root; x = next; }
        return root;
    }
    template<class E> void union_set(E x, E y) { link(find_set(x), find_set(y)); }
    template<class E> void link(E i, E j)
    {
        if(i == j) return;
        if(rank[i] > rank[j]) {
            parent[j] = i;
        } else {
            parent[i] = j;
            if(rank[i] == rank[j]) ++rank[j];
        }
    }
private:
    RankPA rank;
    ParentPA parent;
};
}
#endif

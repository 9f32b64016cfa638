// TEST INFRASTRUCTURE ONLY.
// Minimal stand-in for <seqan/align.h> (SeqAn 2.4.0 is not installed in this
// container) so that /root/reference/src/Align4.cpp can be compiled IN PLACE,
// unmodified, into oracle/_ref/ (and src/AssemblerAlign3.cpp, see ../ref_align3.cpp).  It
// provides exactly the names used at src/Align4.cpp:1001-1043 and src/AssemblerAlign3.cpp:42-260.  The dynamic programming itself is delegated to
// oracle/banded_dp.hpp (restated algorithm; tie policy UNPINNED, see there).
// Every other line of Align4 executed by oracle/_ref is the reference's own.
#ifndef SHIM_SEQAN_ALIGN_H
#define SHIM_SEQAN_ALIGN_H

#include <cstdint>
#include <limits>
#include <vector>
#include "../../../banded_dp.hpp"

namespace seqan {

template<class T> class String : public std::vector<T> {};
template<class T, class V> inline void appendValue(String<T>& s, const V& v) { s.push_back(T(v)); }
template<class T> inline std::size_t length(const String<T>& s) { return s.size(); }

struct OwnerTag {};
template<class X = void> struct Dependent {};
template<class TSeq, class TSpec = OwnerTag> class StringSet : public std::vector<TSeq> {};
template<class TSeq, class TSpec> inline void appendValue(StringSet<TSeq, TSpec>& s, const TSeq& v) { s.push_back(v); }

template<class TStringSet> struct Alignment {};
template<class TSpec> class Graph;

// Graph<Alignment<StringSet<String<T>, Dependent<>>>>: holds the two sequences
// and, after globalAlignment, the DP result.
template<class T>
class Graph< Alignment< StringSet< String<T>, Dependent<> > > > {
public:
    String<T> seq0, seq1;
    oracle::BandedDpResult dp;
    template<class TSet> explicit Graph(const TSet& s) : seq0(s[0]), seq1(s[1]) {}
};

struct Simple {};
template<class TValue, class TSpec> class Score;
template<class TValue> class Score<TValue, Simple> {
public:
    TValue match, mismatch, gap;
    Score(TValue match, TValue mismatch, TValue gap) : match(match), mismatch(mismatch), gap(gap) {}
};

template<bool TTop, bool TLeft, bool TRight, bool TBottom> struct AlignConfig {};
struct LinearGaps {};
template<class T> struct MinValue { static constexpr T VALUE = std::numeric_limits<T>::min(); };

// Only the all-free-end-gaps configuration is used by Align4.
template<class T>
inline int globalAlignment(
    Graph< Alignment< StringSet< String<T>, Dependent<> > > >& g,
    const Score<int, Simple>& score,
    AlignConfig<true, true, true, true>,
    int lowerDiagonal, int upperDiagonal,
    LinearGaps)
{
    oracle::bandedOverlapAlignment(
        g.seq0.data(), uint32_t(g.seq0.size()),
        g.seq1.data(), uint32_t(g.seq1.size()),
        score.match, score.mismatch, score.gap,
        lowerDiagonal, upperDiagonal, g.dp);
    return g.dp.ok ? g.dp.score : MinValue<int>::VALUE;
}

// Unbanded overload (step 1 of align method 3, src/AssemblerAlign3.cpp:118-122): the same
// recurrence over the whole matrix, i.e. every diagonal -ny .. nx is inside the band.
template<class T>
inline int globalAlignment(
    Graph< Alignment< StringSet< String<T>, Dependent<> > > >& g,
    const Score<int, Simple>& score,
    AlignConfig<true, true, true, true> config,
    LinearGaps gaps)
{
    return globalAlignment(g, score, config, -int(g.seq1.size()), int(g.seq0.size()), gaps);
}

// Two gapped rows, concatenated; gap symbol is '-' == 45 (src/Align4.cpp:1007).
template<class T>
inline void convertAlignment(
    const Graph< Alignment< StringSet< String<T>, Dependent<> > > >& g,
    String<T>& align)
{
    const T gap = T(45);
    std::vector<T> row0, row1;
    const auto& dp = g.dp;
    const uint32_t nx = uint32_t(g.seq0.size());
    const uint32_t ny = uint32_t(g.seq1.size());
    uint32_t x = 0, y = 0;
    // Leading free gaps.
    for(; x < dp.iBegin; x++) { row0.push_back(g.seq0[x]); row1.push_back(gap); }
    for(; y < dp.jBegin; y++) { row0.push_back(gap); row1.push_back(g.seq1[y]); }
    for(const uint8_t op : dp.ops) {
        if(op == oracle::TRACE_DIAG)      { row0.push_back(g.seq0[x++]); row1.push_back(g.seq1[y++]); }
        else if(op == oracle::TRACE_VERT) { row0.push_back(gap);         row1.push_back(g.seq1[y++]); }
        else                              { row0.push_back(g.seq0[x++]); row1.push_back(gap); }
    }
    // Trailing free gaps.
    for(; x < nx; x++) { row0.push_back(g.seq0[x]); row1.push_back(gap); }
    for(; y < ny; y++) { row0.push_back(gap); row1.push_back(g.seq1[y]); }
    align.clear();
    align.insert(align.end(), row0.begin(), row0.end());
    align.insert(align.end(), row1.begin(), row1.end());
}

}  // namespace seqan
#endif

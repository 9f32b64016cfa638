// Assembler::suppressAlignmentCandidates (src/AssemblerAlign.cpp:1168-1240): the host step the human Nanopore
// configurations run between the two seams (srcMain/main.cpp:697-702; conf/Nanopore-May2022.conf sets
// Align.sameChannelReadAlignment.suppressDeltaThreshold = 30).  A candidate is dropped when its two reads come
// from the same channel, sample and run and their `read=` numbers differ by less than delta
// (Assembler::suppressAlignment, :1078-1162): consecutive reads of one pore are the two strands of one molecule
// more often than an overlap.  String metadata, one pass over the candidates: host work.
#include "OverlapStages.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <vector>

namespace shasta_mi355x {
namespace host {

namespace {

struct Text { const char* begin; const char* end; bool empty() const { return begin == end; } size_t size() const { return size_t(end - begin); } };

bool same(const Text& a, const Text& b) { return a.size() == b.size() && std::equal(a.begin, a.end, b.begin); }

// Reads::getMetaData (src/Reads.cpp:264-320): the meta data is a whitespace-separated list of Key=Value
// tokens; the value of the first token that is longer than "key=" and starts with it, else empty.
Text metaDataValue(const char* begin, const char* end, const std::string& key)
{
    const char* p = begin;
    while(p != end) {
        const char* q = p;
        while(q != end && !std::isspace(static_cast<unsigned char>(*q))) ++q;
        if(q > p + key.size() + 1 && std::equal(key.begin(), key.end(), p) && p[key.size()] == '=') return Text{p + key.size() + 1, q};
        if(q == end) break;
        p = q;
        while(p != end && std::isspace(static_cast<unsigned char>(*p))) ++p;
    }
    return Text{end, end};
}

// atoul of src/span.hpp:64-80: decimal digits only (anything else throws), arithmetic modulo 2^64.
uint64_t decimal(const Text& t)
{
    uint64_t n = 0;
    for(const char* p = t.begin; p != t.end; ++p) {
        if(!std::isdigit(static_cast<unsigned char>(*p))) throw std::runtime_error("Non-digit found in " + std::string(t.begin, t.end));
        n = n * 10 + uint64_t(*p - '0');
    }
    return n;
}

// Assembler::suppressAlignment (src/AssemblerAlign.cpp:1078-1162) on the meta data of two reads.
bool suppressPair(const char* m0, const char* e0, const char* m1, const char* e1, uint64_t delta)
{
    for(const char* key : {"ch", "sampleid", "runid"}) {                               // :1089-1131
        const Text v0 = metaDataValue(m0, e0, key);
        if(v0.empty()) return false;
        const Text v1 = metaDataValue(m1, e1, key);
        if(v1.empty()) return false;
        if(!same(v0, v1)) return false;
    }
    const Text read0 = metaDataValue(m0, e0, "read");                                  // :1138-1146
    if(read0.empty()) return false;
    const Text read1 = metaDataValue(m1, e1, "read");
    if(read1.empty()) return false;
    const int64_t r0 = int64_t(decimal(read0)), r1 = int64_t(decimal(read1));       // :1152-1153
    return std::llabs(r0 - r1) < int64_t(delta);                                       // :1160
}

}  // namespace

// The same step on arrays in memory (what a caller that holds the candidates of the first seam in memory runs before the second
// -- bench.py's steps of the configs[3] / [4] workloads): candidates compacted in place, no side file, no console lines.
uint64_t suppressAlignmentCandidatesInMemory(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount,
    shasta_oriented_read_pair* candidates, uint64_t candidateCount, uint64_t delta)
{
    uint64_t kept = 0;
    for(uint64_t i = 0; i < candidateCount; i++) {
        const shasta_oriented_read_pair c = candidates[i];
        if(c.readIds[0] >= readCount || c.readIds[1] >= readCount) throw std::runtime_error("suppressAlignmentCandidates: a candidate names a read that has no meta data.");
        const bool drop = suppressPair(metaData + metaDataToc[c.readIds[0]], metaData + metaDataToc[c.readIds[0] + 1],
            metaData + metaDataToc[c.readIds[1]], metaData + metaDataToc[c.readIds[1] + 1], delta);
        if(!drop) candidates[kept++] = c;
    }
    return kept;
}

uint64_t suppressAlignmentCandidates(const std::string& dataDirectory, uint64_t delta, size_t /* threadCount */)
{
    using Strings = MappedVectorOfVectors<char, uint64_t>;
    Strings names, metaData;
    names.accessExistingReadOnly(dataDirectory + "/ReadNames");
    metaData.accessExistingReadOnly(dataDirectory + "/ReadMetaData");
    AlignmentCandidates candidates;
    candidates.accessExistingReadWrite(dataDirectory + "/AlignmentCandidates");
    const uint64_t candidateCount = candidates.size();
    const uint64_t readCount = metaData.size();

    auto suppress = [&](uint32_t readId0, uint32_t readId1) {
        if(readId0 >= readCount || readId1 >= readCount) throw std::runtime_error("suppressAlignmentCandidates: a candidate names a read that has no meta data.");
        return suppressPair(metaData.begin(readId0), metaData.begin(readId0) + (metaData.toc[readId0 + 1] - metaData.toc[readId0]),
            metaData.begin(readId1), metaData.begin(readId1) + (metaData.toc[readId1 + 1] - metaData.toc[readId1]), delta);
    };

    std::ofstream csv("SuppressedAlignmentCandidates.csv");                                 // :1187-1188
    csv << "ReadId0,ReadId1,SameStrand,Name0,Name1,MetaData0,MetaData1" << std::endl;
    std::cout << "Number of alignment candidates before suppression is " << candidateCount << std::endl;
    auto text = [](const Strings& v, uint32_t i) { return std::string(v.begin(i), v.begin(i) + (v.toc[i + 1] - v.toc[i])); };
    uint64_t kept = 0, suppressCount = 0;
    for(uint64_t i = 0; i < candidateCount; i++) {
        const shasta_oriented_read_pair c = candidates[i];
        if(suppress(c.readIds[0], c.readIds[1])) {
            ++suppressCount;
            csv << c.readIds[0] << "," << c.readIds[1] << "," << (c.isSameStrand ? "Yes" : "No") << ","
                << text(names, c.readIds[0]) << "," << text(names, c.readIds[1]) << ","
                << text(metaData, c.readIds[0]) << "," << text(metaData, c.readIds[1]) << std::endl;
        } else {
            candidates[kept++] = c;
        }
    }
    candidates.resize(kept);                                                                // :1210
    std::cout << "Suppressed " << suppressCount << " alignment candidates." << std::endl;
    std::cout << "Number of alignment candidates after suppression is " << kept << std::endl;
    return suppressCount;
}

}  // namespace host
}  // namespace shasta_mi355x

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
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
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

// The decision from keys made once per read: the values of ch / sampleid / runid as numbers (equal strings, equal numbers; 0: no value)
// and the read number.  The reference parses the two reads' meta data anew for every candidate (string searches: 0.2 us a candidate,
// half a second for the 2.4 million candidates of 100 000 reads); a read's meta data does not change between candidates.
void suppressionKeys(const uint64_t* metaDataToc, const char* metaData, uint64_t readCount, SuppressionKey* keys)
{
    std::unordered_map<std::string, uint32_t> ids[3];
    const char* const names[3] = {"ch", "sampleid", "runid"};
    for(uint64_t r = 0; r < readCount; r++) {
        const char* const begin = metaData + metaDataToc[r];
        const char* const end = metaData + metaDataToc[r + 1];
        SuppressionKey& k = keys[r];
        for(int f = 0; f < 3; f++) {
            const Text v = metaDataValue(begin, end, names[f]);
            k.field[f] = 0;
            if(!v.empty()) k.field[f] = ids[f].emplace(std::string(v.begin, v.end), uint32_t(ids[f].size() + 1)).first->second;
        }
        const Text read = metaDataValue(begin, end, "read");
        k.flags = 0; k.read = 0;
        if(!read.empty()) {
            k.flags = 1;
            // (a value that is not a number makes the reference throw when a candidate gets as far as reading it, not before)
            for(const char* p = read.begin; p != read.end; ++p) if(!std::isdigit(static_cast<unsigned char>(*p))) k.flags = 3;
            if(k.flags == 1) k.read = decimal(read);
        }
    }
}

uint64_t suppressAlignmentCandidatesByKeys(const SuppressionKey* keys, uint64_t readCount, const shasta_oriented_read_pair* candidates, uint64_t candidateCount,
    shasta_oriented_read_pair* out, uint64_t delta, size_t threadCount)
{
    auto drop = [&](const shasta_oriented_read_pair& c) {
        if(c.readIds[0] >= readCount || c.readIds[1] >= readCount) throw std::runtime_error("suppressAlignmentCandidates: a candidate names a read that has no meta data.");
        const SuppressionKey& a = keys[c.readIds[0]];
        const SuppressionKey& b = keys[c.readIds[1]];
        for(int f = 0; f < 3; f++) if(a.field[f] == 0 || b.field[f] == 0 || a.field[f] != b.field[f]) return false;      // :1089-1131
        if(!(a.flags & 1)) return false;                                                                             // :1138-1146
        if(a.flags & 2) throw std::runtime_error("Non-digit found in the read number of a read's meta data");
        if(!(b.flags & 1)) return false;
        if(b.flags & 2) throw std::runtime_error("Non-digit found in the read number of a read's meta data");
        return std::llabs(int64_t(a.read) - int64_t(b.read)) < int64_t(delta);                                       // :1152-1160
    };
    // Slices on as many threads: each lists the candidates it drops (few), the slices' places in the output follow from the counts, and
    // each then copies the runs between its dropped candidates to its place -- `out` may be `candidates` itself (one thread then: the
    // runs move down in order) or another array (the copy a caller needs anyway, done here once and in parallel).
    const bool inPlace = out == candidates;
    const size_t threads = inPlace ? 1 : std::max<size_t>(1, std::min<size_t>(threadCount ? threadCount : 8, candidateCount / 65536 + 1));
    std::vector<uint64_t> begin(threads + 1), base(threads + 1, 0);
    for(size_t t = 0; t <= threads; t++) begin[t] = candidateCount * t / threads;
    std::vector<std::vector<uint64_t>> dropped(threads);
    std::vector<std::string> errors(threads);
    auto onThreads = [&](auto&& f) {
        std::vector<std::thread> others;
        for(size_t t = 1; t < threads; t++) others.emplace_back(f, t);
        f(0);
        for(std::thread& t : others) t.join();
        for(const std::string& e : errors) if(!e.empty()) throw std::runtime_error(e);
    };
    onThreads([&](size_t t) {
        try { for(uint64_t i = begin[t]; i < begin[t + 1]; i++) if(drop(candidates[i])) dropped[t].push_back(i); }
        catch(const std::exception& e) { errors[t] = e.what(); }
    });
    for(size_t t = 0; t < threads; t++) base[t + 1] = base[t] + (begin[t + 1] - begin[t]) - dropped[t].size();
    onThreads([&](size_t t) {
        uint64_t from = begin[t], to = base[t];
        auto run = [&](uint64_t end) {
            if(end > from && (!inPlace || to != from)) std::memmove(out + to, candidates + from, (end - from) * sizeof(shasta_oriented_read_pair));
            to += end - from;
        };
        for(const uint64_t i : dropped[t]) { run(i); from = i + 1; }
        run(begin[t + 1]);
    });
    return base[threads];
}

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

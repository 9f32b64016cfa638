// shasta_mi355x_stage: the stage-level entry points of the reference's scripts
// (scripts/FindAlignmentCandidatesLowHash0.py, scripts/ComputeAlignments.py) on an existing
// Data/ directory, with the work done on the GPU.
//   shasta_mi355x_stage lowhash0 <Data> [m hashFraction minHashIterationCount alignmentCandidatesPerRead
//                                        log2MinHashBucketCount minBucketSize maxBucketSize minFrequency]
//   shasta_mi355x_stage palindromic <Data> [maxSkip maxDrift maxMarkerFrequency alignedFractionThreshold nearDiagonalFractionThreshold deltaThreshold]
//   shasta_mi355x_stage suppress <Data> [delta]
//   shasta_mi355x_stage candidate-table <Data>
//   shasta_mi355x_stage read-graph <Data> [maxAlignmentCount maxTrim]
//   shasta_mi355x_stage align    <Data> [minAlignedMarkerCount minAlignedFraction maxSkip maxDrift maxTrim suppressContainments]
// Exit codes follow srcMain/main.cpp:103-129: 0 success, 1 std::runtime_error / other exception.
#include "OverlapStages.hpp"
#include "PalindromicReads.hpp"

#include <cstdlib>
#include <iostream>
#include <string>

using namespace shasta_mi355x::host;

int main(int argc, char** argv)
{
    try {
        if(argc < 3) throw std::runtime_error("usage: shasta_mi355x_stage markers|palindromic|lowhash0|suppress|candidate-table|align|read-graph <DataDirectory> [options...]");
        const std::string command = argv[1], data = argv[2];
        auto arg = [&](int k, const char* fallback) { return std::string(argc > k ? argv[k] : fallback); };
        if(command == "lowhash0") {
            // Defaults: MinHashOptions, src/AssemblerOptions.cpp:327-371.
            findAlignmentCandidatesLowHash0(data,
                std::stoull(arg(3, "4")), std::stod(arg(4, "0.01")), std::stoull(arg(5, "10")), std::stod(arg(6, "20")),
                std::stoull(arg(7, "0")), std::stoull(arg(8, "0")), std::stoull(arg(9, "10")), std::stoull(arg(10, "2")), 0);
        } else if(command == "markers") {
            // Assembler::findMarkers (srcMain/main.cpp: the step before the two seams).
            findMarkers(data, 0);
        } else if(command == "palindromic") {
            // Assembler::flagPalindromicReads, srcMain/main.cpp:654-663, the step before LowHash0; defaults src/AssemblerOptions.cpp:255-288.
            PalindromicReadOptions o;
            o.maxSkip = uint32_t(std::stoul(arg(3, "100"))); o.maxDrift = uint32_t(std::stoul(arg(4, "100")));
            o.maxMarkerFrequency = uint32_t(std::stoul(arg(5, "10")));
            o.alignedFractionThreshold = std::stod(arg(6, "0.1")); o.nearDiagonalFractionThreshold = std::stod(arg(7, "0.1"));
            o.deltaThreshold = uint32_t(std::stoul(arg(8, "100")));
            (void)flagPalindromicReads(data, o, 0);
        } else if(command == "suppress") {
            // Assembler::suppressAlignmentCandidates(delta), srcMain/main.cpp:697-702 (run when delta > 0).
            (void)suppressAlignmentCandidates(data, std::stoull(arg(3, "30")), 0);
        } else if(command == "candidate-table") {
            // Assembler::computeCandidateTable, between the two seams (srcMain/main.cpp:706).
            Markers markers;
            markers.accessExistingReadOnly(data + "/Markers");
            AlignmentCandidates candidates;
            candidates.accessExistingReadOnly(data + "/AlignmentCandidates");
            computeCandidateTable(markers.size() / 2, candidates, data);
        } else if(command == "align") {
            AlignOptions o;
            o.minAlignedMarkerCount = std::stoull(arg(3, "100")); o.minAlignedFraction = std::stod(arg(4, "0"));
            o.maxSkip = std::stoull(arg(5, "30")); o.maxDrift = std::stoull(arg(6, "30")); o.maxTrim = std::stoull(arg(7, "30"));
            o.suppressContainments = std::stoull(arg(8, "0")) != 0;
            o.alignMethod = std::stoi(arg(9, "4"));                       // 3 needs Data/Kmers (for k)
            o.downsamplingFactor = std::stod(arg(10, "0.1")); o.bandExtend = std::stoi(arg(11, "10"));
            computeAlignments(data, o, 0);
        } else if(command == "read-graph") {
            // Assembler::createReadGraph(maxAlignmentCount, maxTrim), srcMain/main.cpp:718-722 (ReadGraph.creationMethod 0).
            createReadGraph(data, uint32_t(std::stoul(arg(3, "6"))), uint32_t(std::stoul(arg(4, "30"))));
        } else {
            throw std::runtime_error("unknown command " + command);
        }
        return 0;
    } catch(const std::exception& e) {
        std::cout << e.what() << std::endl;
        return 1;
    }
}

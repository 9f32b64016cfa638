// SeqAn pin kit -- TEST INFRASTRUCTURE ONLY.  The one program of this repository that is meant to be compiled against the
// REAL SeqAn 2.4.0 (Ubuntu: apt install libseqan2-dev; docs/Prerequisites.html:110 of the reference), on any machine that has it:
//
//     g++ -std=c++14 -O2 pin_seqan.cpp -o pin_seqan && ./pin_seqan cases.txt | python3 which_policy.py
//
// For every case of cases.txt it makes exactly the call of /root/reference/src/Align4.cpp:1001-1043 (String<KmerId> sequences
// with 100 added to every id, Score<int, Simple>(6, -1, -1), AlignConfig<true, true, true, true>, the band, LinearGaps) and
// then the reference's own loop over convertAlignment's two rows (:1041-1068), and prints the score and the aligned marker
// ordinals.  which_policy.py compares the lines with what each of the 12 tie policies of oracle/banded_dp.hpp gives.
#include <seqan/align.h>

#include <cstdint>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    using namespace seqan;
    using KmerId = uint32_t;
    using TSequence = String<KmerId>;
    using TStringSet = StringSet<TSequence>;
    using TDepStringSet = StringSet<TSequence, Dependent<> >;
    using TAlignGraph = Graph<seqan::Alignment<TDepStringSet> >;
    const uint32_t seqanGapValue = 45;
    if(argc != 2) { std::cerr << "usage: pin_seqan cases.txt" << std::endl; return 2; }
    std::ifstream in(argv[1]);
    std::string line;
    while(std::getline(in, line)) {
        if(line.empty() || line[0] == '#') continue;
        for(char& c : line) if(c == '|') c = '\n';
        std::istringstream parts(line);
        std::string head, text0, text1, name;
        std::getline(parts, head); std::getline(parts, text0); std::getline(parts, text1);
        int bandMin = 0, bandMax = 0;
        std::istringstream(head) >> name >> bandMin >> bandMax;
        std::vector<KmerId> markers[2];
        { std::istringstream s(text0); KmerId v; while(s >> v) markers[0].push_back(v); }
        { std::istringstream s(text1); KmerId v; while(s >> v) markers[1].push_back(v); }
        const uint32_t nx = uint32_t(markers[0].size()), ny = uint32_t(markers[1].size());

        TSequence sequences[2];
        for(int i = 0; i < 2; i++) for(const KmerId kmerId : markers[i]) appendValue(sequences[i], kmerId + 100);
        TStringSet sequencesSet;
        appendValue(sequencesSet, sequences[0]);
        appendValue(sequencesSet, sequences[1]);
        TAlignGraph graph(sequencesSet);
        const int score = globalAlignment(graph, Score<int, Simple>(6, -1, -1), AlignConfig<true, true, true, true>(), bandMin, bandMax, LinearGaps());
        std::cout << name << " score " << score << " pairs";
        if(score != MinValue<int>::VALUE) {
            TSequence align;
            convertAlignment(graph, align);
            const int alignmentLength = int(length(align)) / 2;
            uint32_t ordinal0 = 0, ordinal1 = 0;
            for(int i = 0; i < alignmentLength && ordinal0 < nx && ordinal1 < ny; i++) {
                if(align[i] != seqanGapValue && align[i + alignmentLength] != seqanGapValue && markers[0][ordinal0] == markers[1][ordinal1]) {
                    std::cout << " " << ordinal0 << ":" << ordinal1;
                }
                if(align[i] != seqanGapValue) ++ordinal0;
                if(align[i + alignmentLength] != seqanGapValue) ++ordinal1;
            }
        }
        std::cout << std::endl;
    }
    return 0;
}

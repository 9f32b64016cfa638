// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/libshasta_ref.so.
//
// Same-channel candidate suppression, the host step the human Nanopore configurations run between the two seams
// (srcMain/main.cpp:697-702, Align.sameChannelReadAlignment.suppressDeltaThreshold = 30).  The reads -- names and
// meta data as the reference stores them -- are loaded from a FASTA file by the reference's own ReadLoader into
// a Reads object whose files live in a Data/ directory (so that other code can read what the reference wrote),
// and the meta data is parsed by the reference's own Reads::getMetaData (src/Reads.cpp:264-320) and atoul
// (src/span.hpp:64-80).  Restated here: the decision Assembler::suppressAlignment (src/AssemblerAlign.cpp:1078-1162),
// which is a member of the Assembler class that cannot be compiled in this container.
#include "ReadLoader.hpp"
#include "Reads.hpp"
#include "span.hpp"

#include <cstdlib>
#include <cstring>
#include <string>

using namespace shasta;

namespace { std::string suppressError; }

extern "C" {

const char* ref_suppress_last_error() { return suppressError.c_str(); }

// Creates Reads, ReadNames, ReadMetaData, ReadFlags ... under dataDirectory from the FASTA file, then
// suppress[i] = Assembler::suppressAlignment(readId0[i], readId1[i], delta).  Returns the number of reads.
int ref_suppress_alignment_flags(const char* fastaPath, const char* dataDirectory, uint64_t candidateCount,
    const uint32_t* readId0, const uint32_t* readId1, uint64_t delta, uint8_t* suppress, uint64_t* readCountOut)
{
    try {
        const std::string d(dataDirectory);
        Reads reads;
        reads.createNew(1, d + "/Reads", d + "/ReadNames", d + "/ReadMetaData", d + "/ReadRepeatCounts", d + "/ReadFlags",
            d + "/ReadIdsSortedByName", 4096);
        {
            ReadLoader loader(fastaPath, 1, 0, false, 1, d + "/", 4096, reads);
        }
        *readCountOut = reads.readCount();
        for(uint64_t i = 0; i < candidateCount; i++) {
            const ReadId r0 = readId0[i], r1 = readId1[i];
            bool s = true;
            for(const char* key : {"ch", "sampleid", "runid"}) {          // :1089-1131, in this order
                const auto v0 = reads.getMetaData(r0, key);
                if(v0.empty()) { s = false; break; }
                const auto v1 = reads.getMetaData(r1, key);
                if(v1.empty()) { s = false; break; }
                if(v0 not_eq v1) { s = false; break; }
            }
            if(s) {
                const auto read0 = reads.getMetaData(r0, "read");       // :1138-1146
                const auto read1 = reads.getMetaData(r1, "read");
                if(read0.empty() || read1.empty()) s = false;
                else {
                    const int64_t n0 = int64_t(atoul(read0)), n1 = int64_t(atoul(read1));     // :1152-1153
                    s = std::abs(n0 - n1) < int64_t(delta);                                    // :1160
                }
            }
            suppress[i] = s ? 1 : 0;
        }
        return 0;
    } catch(std::exception& e) { suppressError = e.what(); return 1; }
}

}  // extern "C"

// TEST INFRASTRUCTURE ONLY.
// Stand-in for the reference's Assembler facade (src/Assembler.hpp, which needs Boost): a class
// of the same name with just the two data members Assembler::alignOrientedReads3 touches, so that
// /root/reference/src/AssemblerAlign3.cpp compiles in place and unmodified (see ref_align3.cpp).
#ifndef SHASTA_ASSEMBLER_HPP
#define SHASTA_ASSEMBLER_HPP

#include "Alignment.hpp"
#include "Kmer.hpp"
#include "Marker.hpp"
#include "MemoryMappedVector.hpp"
#include "MemoryMappedVectorOfVectors.hpp"
#include "ReadId.hpp"
#include "SHASTA_ASSERT.hpp"
#include "algorithm.hpp"
#include "array.hpp"
#include "iostream.hpp"
#include "span.hpp"
#include "stdexcept.hpp"
#include "string.hpp"
#include "utility.hpp"
#include "vector.hpp"

namespace shasta {
    class Assembler {
    public:
        MemoryMapped::VectorOfVectors<CompressedMarker, uint64_t> markers;      // Data/Markers
        MemoryMapped::Vector<KmerInfo> kmerTable;                                // Data/Kmers
        void alignOrientedReads3(
            OrientedReadId, OrientedReadId,
            int matchScore, int mismatchScore, int gapScore,
            double downsamplingFactor, int bandExtend, int maxBand,
            Alignment&, AlignmentInfo&);
    };
}
#endif

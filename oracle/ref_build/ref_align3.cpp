// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/libshasta_ref.so.
//
// Compiles the reference's /root/reference/src/AssemblerAlign3.cpp IN PLACE and unmodified
// (Assembler::alignOrientedReads3, the whole of align method 3).  That file includes
// "Assembler.hpp", which drags in Boost: ref_facade.hpp claims its include guard first.
// SeqAn is shims/seqan/align.h (restated DP, tie policy UNPINNED); PngImage is stubbed in
// ref_shims.cpp (only reached with debug=true).
#include "ref_facade.hpp"
#include "AssemblerAlign3.cpp"

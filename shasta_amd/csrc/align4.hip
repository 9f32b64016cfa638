// placeholder, replaced below
#include "context.hpp"
namespace shasta_mi355x {
void align4Run(Context&, uint64_t, const shasta_oriented_read_pair*, const shasta_align4_options&, bool, shasta_align4_result&)
{ throw std::runtime_error("align4: not built yet"); }
void align4Free(shasta_align4_result&) {}
void bandedDpUnit(const uint32_t*, uint32_t, const uint32_t*, uint32_t, int32_t, int32_t, uint32_t*, uint64_t, uint64_t*, int32_t*)
{ throw std::runtime_error("banded_dp: not built yet"); }
}

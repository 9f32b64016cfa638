// TEST INFRASTRUCTURE ONLY (force-included in front of align4.hip by tests/emu/Makefile).
// align4CellsChunkKernel declares its dynamic LDS as `extern __shared__ uint32_t ldsWords[];`
// (align3WideDpKernel: `extern __shared__ int32_t wideRows[];`);
// under emulation that block-scope extern refers to this definition: the largest allocation a
// workgroup of gfx950 can ask for (160 KB), one copy per OS thread = per resident workgroup.
#pragma once
#include <cstdint>
namespace shasta_mi355x { namespace { thread_local uint32_t ldsWords[160 * 1024 / 4]; thread_local int32_t wideRows[160 * 1024 / 4]; } }

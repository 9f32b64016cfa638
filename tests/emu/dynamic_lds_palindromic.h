// TEST INFRASTRUCTURE ONLY (force-included in front of palindromic.hip by tests/emu/Makefile).
// palindromicScreenKernel declares its dynamic LDS as `extern __shared__ uint32_t screenWindows[];`;
// under emulation that block-scope extern refers to this definition (one copy per OS thread = per
// resident workgroup), sized for the largest window the host code accepts (4 wavefronts x (64 + 2 x 4096) words).
#pragma once
#include <cstdint>
namespace shasta_mi355x { namespace { thread_local uint32_t screenWindows[4 * (64 + 2 * 4096)]; } }

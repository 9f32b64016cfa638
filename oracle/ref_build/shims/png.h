// TEST INFRASTRUCTURE ONLY.  Stand-in for <png.h>: Align4's PNG output is
// debug-only (src/Align4.cpp:151-161); PngImage is stubbed in ref_shims.cpp.
#ifndef SHIM_PNG_H
#define SHIM_PNG_H
typedef unsigned char png_byte;
#endif

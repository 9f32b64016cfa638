// TEST INFRASTRUCTURE ONLY.
// The two symbols the reference TU closure needs that cannot be compiled here:
//  - shasta::timestamp (src/timestamp.cpp uses boost::date_time)
//  - shasta::PngImage  (src/PngImage.cpp uses libpng; only reached with debug=true)
#include "timestamp.hpp"
#include "PngImage.hpp"
#include <stdexcept>
std::ostream& shasta::timestamp(std::ostream& s) { return s; }
shasta::PngImage::PngImage(int width, int height) : width(width), height(height) {}
void shasta::PngImage::setPixel(int, int, int, int, int) {}
void shasta::PngImage::write(const string&) const { throw std::runtime_error("PngImage is stubbed in oracle/_ref"); }
void shasta::PngImage::writeGrid(int, int, int, int) {}
shasta::PngImage::PngImage(const PngImage& that, int) : width(that.width), height(that.height) {}   // AlignmentGraph::writeImage (debug only)
void shasta::PngImage::magnify(int) {}

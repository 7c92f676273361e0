// gray_io.h -- 8-bit gray image files as the reference's command-line tool reads and writes them
// (reference: src/FileIO.c:81-159 PGM "P5", :170-287 8-bit BMP).  Host-only.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace nblic {

struct GrayImage {
    int h = 0, w = 0;
    std::vector<uint8_t> px;      // row-major, top-down, stride == w
};

enum class GrayFormat { kNone = 0, kPgm = 1, kBmp = 2 };

// Binary PGM: "P5", then width, height, maxval as decimal integers separated by white space (no
// comment lines, as the reference's fscanf-based reader), maxval in [1,255], ONE white-space byte, then
// w*h bytes.  Samples are taken as they are (no rescaling for maxval < 255).
bool read_pgm(const std::string &path, GrayImage &img);
// Uncompressed 8-bit BMP with one colour plane: pixel = palette INDEX (the palette itself is not read),
// rows stored bottom-up and padded to a multiple of four bytes; negative (top-down) heights are refused.
bool read_bmp8(const std::string &path, GrayImage &img);
// The probing order of the reference's front end: PGM first, then BMP, whatever the file is called
// (NBLIC_main.c:168-169).
GrayFormat read_gray(const std::string &path, GrayImage &img);

// "P5\n<w> <h>\n255\n" + pixels (FileIO.c:141-159)
bool write_pgm(const std::string &path, const uint8_t *px, int h, int w);
// 14-byte file header + 40-byte BITMAPINFOHEADER + 256-entry gray palette (B, G, R, 0xFF) = 1078 bytes,
// then the rows bottom-up, each padded with zeros to a multiple of four (FileIO.c:229-287)
bool write_bmp8(const std::string &path, const uint8_t *px, int h, int w);

bool read_file(const std::string &path, std::vector<uint8_t> &bytes, size_t limit);
bool write_file(const std::string &path, const uint8_t *bytes, size_t n);

}  // namespace nblic

// cli.cpp -- command-line front end with the reference tool's grammar (src/NBLIC_main.c:52-254),
// on top of this library's drop-in entry points.  Exported as nblic_amd_cli_main so that the
// `nblic_codec_amd` executable is a three-line main and tests can drive it in-process.
//
// Grammar kept from the reference: every argument that starts with '-' is a GROUP of one-letter
// switches ("-cn2e2V" == "-c -n2 -e2 -V"): c/C compress, d/D decompress, v verbose, V verbose with
// progress, n<digits> near, e<digit> effort (the character after 'e' is consumed either way), t/T
// the multithread request of effort 0; unknown letters are ignored.  The first other argument is
// the input file, the LAST other argument the output file.  Defaults: compress, -n0 -e1.
// Compress: the input is probed as PGM first, then as BMP, whatever its name; -n0 -e0 selects QNBLIC
// (whose length in 16-bit words is doubled), anything else NBLIC (which clamps near and effort).
// Decompress: QNBLIC is tried first, then NBLIC; the output is a BMP iff its name ends in ".bmp"
// (any case), else a PGM.  Exit status 0, or -1 (255) after a "***Error" line.
// Additions (letters the reference ignores): L lifts the 100,000,000-pixel limit for this run
// (nblic_amd_set_max_pixels), g<digits> selects the HIP device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/nblic_amd.h"
#include "gray_io.h"

namespace nblic {

struct CliOptions {
    std::string src, dst;
    int decompress = 0, near = 0, effort = 1, verbose = 0, multithread = 0, large = 0, device = -1;
    bool have_src = false, have_dst = false;
};

static void parse_switch_group(const char *a, CliOptions &o) {
    for (; *a; a++) {
        switch (*a) {
            case 'c': case 'C': o.decompress = 0; break;
            case 'd': case 'D': o.decompress = 1; break;
            case 'v': o.verbose = 1; break;
            case 'V': o.verbose = 2; break;
            case 'n': case 'N':
                o.near = 0;
                for (; a[1] >= '0' && a[1] <= '9'; a++) o.near = o.near * 10 + (a[1] - '0');
                break;
            case 'e': case 'E':
                if (a[1] >= '0' && a[1] <= '9') o.effort = a[1] - '0';
                if (a[1]) a++;                                   // the reference steps over the next character regardless (NBLIC_main.c:84)
                break;
            case 't': case 'T': o.multithread = 1; break;
            case 'L': o.large = 1; break;
            case 'g':
                o.device = 0;
                for (; a[1] >= '0' && a[1] <= '9'; a++) o.device = o.device * 10 + (a[1] - '0');
                break;
            default: break;
        }
    }
}

CliOptions parse_command(int argc, char **argv) {
    CliOptions o;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] == '-') parse_switch_group(a + 1, o);
        else if (!o.have_src) { o.src = a; o.have_src = true; }
        else { o.dst = a; o.have_dst = true; }
    }
    return o;
}

static bool ends_with_nocase(const std::string &s, const char *suffix) {
    const size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t k = 0; k < n; k++) {
        char a = s[s.size() - n + k], b = suffix[k];
        if (a >= 'A' && a <= 'Z') a = char(a + 32);
        if (b >= 'A' && b <= 'Z') b = char(b + 32);
        if (a != b) return false;
    }
    return true;
}

static const char *kUsage =
    "nblic_codec_amd: NBLIC v0.3 lossless / near-lossless 8-bit gray image codec, MI355X (gfx950) build\n"
    "  compress  : nblic_codec_amd -c [-n<near 0..9>] [-e<effort 0..3>] [-v|-V] [-t] <image.pgm|.bmp> <out.nblic>\n"
    "  decompress: nblic_codec_amd -d [-v|-V] <in.nblic> <image.pgm|.bmp>\n"
    "  switches may be grouped (-cn2e2V); -n0 -e0 is the fast QNBLIC mode; near > 0 needs effort >= 1\n"
    "  additions: -L lift the 100,000,000-pixel limit, -g<N> HIP device\n";

static int cli_run(const CliOptions &o) {
    if (!o.have_src || !o.have_dst) { fputs(kUsage, stdout); return -1; }
    if (o.device >= 0) { char v[16]; snprintf(v, sizeof v, "%d", o.device); setenv("NBLIC_AMD_DEVICE", v, 1); }
    if (o.verbose) { printf("  input  file        = %s\n", o.src.c_str()); printf("  output file        = %s\n", o.dst.c_str()); }
    int near = o.near, effort = o.effort, height = -1, width = -1;
    if (!o.decompress) {
        GrayImage img;
        const GrayFormat fmt = read_gray(o.src, img);
        if (fmt == GrayFormat::kNone) {
            printf("  ***Error : open %s failed\n", o.src.c_str());
            printf("             please specific a gray 8-bit PGM or BMP file as input\n");
            return -1;
        }
        height = img.h; width = img.w;
        if (o.verbose) {
            printf("  input image format = %s\n", fmt == GrayFormat::kBmp ? "BMP" : "PGM");
            printf("  input image shape  = %d x %d\n", width, height);
        }
        if (o.large) nblic_amd_set_max_pixels(nullptr, 1L << 33);
        const size_t n = size_t(height) * size_t(width);
        std::vector<uint16_t> buf(n + 4096);                      // the reference provides 2 bytes per pixel (NBLIC_main.c:141)
        long len;
        if (near == 0 && effort == 0) {
            const int words = o.multithread ? QNBLICcompressMultiThread(buf.data(), img.px.data(), height, width)
                                            : QNBLICcompress(buf.data(), img.px.data(), height, width);
            len = words < 0 ? -1 : 2L * words;
        } else {
            len = NBLICcompress(o.verbose > 1, reinterpret_cast<unsigned char *>(buf.data()), img.px.data(), height, width, &near, &effort);
        }
        if (len < 0) { printf("  ***Error : compress failed\n"); return -1; }
        if (o.verbose) {
            printf("  effort             = %d\n", effort);
            printf("  near               = %d (%s)\n", near, near <= 0 ? "lossless" : "lossy");
            printf("  output size        = %ld B\n", len);
            printf("  compression rate   = %.5f\n", (1.0 * width * height) / double(len));
            printf("  compression bpp    = %.5f\n", (8.0 * double(len)) / (double(width) * height));
        }
        if (!write_file(o.dst, reinterpret_cast<const uint8_t *>(buf.data()), size_t(len))) { printf("  ***Error : write %s failed\n", o.dst.c_str()); return -1; }
    } else {
        std::vector<uint8_t> bytes;
        if (!read_file(o.src, bytes, size_t(1) << 33)) { printf("  ***Error : open %s failed\n", o.src.c_str()); return -1; }
        if (o.verbose) printf("  input size         = %zu B\n", bytes.size());
        if (o.large) nblic_amd_set_max_pixels(nullptr, 1L << 33);
        near = 0; effort = 0;
        // plane size from whichever header this is; both decoders validate the rest
        size_t n = 0;
        if (bytes.size() >= 8 && memcmp(bytes.data(), "Q0.2", 4) == 0) n = size_t(bytes[4] | (bytes[5] << 8)) * size_t(bytes[6] | (bytes[7] << 8));
        else if (bytes.size() >= 16 && memcmp(bytes.data(), "NBLIC0.3", 8) == 0) n = size_t((bytes[9] << 8) | bytes[10]) * size_t((bytes[11] << 8) | bytes[12]);
        std::vector<uint8_t> px(n + 16);
        bytes.resize(bytes.size() + 16 + (bytes.size() & 1));     // both ABIs take no length; keep a readable margin, 2-byte units
        int rc = QNBLICdecompress(reinterpret_cast<uint16_t *>(bytes.data()), px.data(), &height, &width);
        if (rc < 0) rc = NBLICdecompress(o.verbose > 1, bytes.data(), px.data(), &height, &width, &near, &effort);
        if (rc < 0) { printf("  ***Error : decompress failed\n"); return -1; }
        const bool bmp = ends_with_nocase(o.dst, ".bmp");
        if (o.verbose) {
            printf("  effort             = %d\n", effort);
            printf("  near               = %d (%s)\n", near, near <= 0 ? "lossless" : "lossy");
            printf("  output image format= %s\n", bmp ? "BMP" : "PGM");
            printf("  output image shape = %d x %d\n", width, height);
        }
        const bool ok = bmp ? write_bmp8(o.dst, px.data(), height, width) : write_pgm(o.dst, px.data(), height, width);
        if (!ok) { printf("  ***Error : write %s failed\n", o.dst.c_str()); return -1; }
    }
    return 0;
}

}  // namespace nblic

extern "C" {

int nblic_amd_cli_main(int argc, char **argv) {
    const int rc = nblic::cli_run(nblic::parse_command(argc, argv));
    fflush(stdout);                                              // callers that embed this (tests) read the text right away
    return rc;
}

// the parsed command line, for tests of the switch grammar: fields[0..7] = decompress, near, effort, verbose,
// multithread, large, device, (have_src | have_dst << 1); the two names are copied into src / dst (cap bytes each)
void nblic_amd_cli_parse(int argc, char **argv, int *fields, char *src, char *dst, size_t cap) {
    const nblic::CliOptions o = nblic::parse_command(argc, argv);
    fields[0] = o.decompress; fields[1] = o.near; fields[2] = o.effort; fields[3] = o.verbose; fields[4] = o.multithread;
    fields[5] = o.large; fields[6] = o.device; fields[7] = int(o.have_src) | (int(o.have_dst) << 1);
    if (cap) { snprintf(src, cap, "%s", o.src.c_str()); snprintf(dst, cap, "%s", o.dst.c_str()); }
}

int nblic_amd_read_gray(const char *path, unsigned char *px, size_t cap, int *h, int *w) {
    nblic::GrayImage img;
    const nblic::GrayFormat f = nblic::read_gray(path, img);
    if (f == nblic::GrayFormat::kNone) return 0;
    *h = img.h; *w = img.w;
    if (img.px.size() > cap) return -1;
    memcpy(px, img.px.data(), img.px.size());
    return int(f);
}

int nblic_amd_write_gray(const char *path, const unsigned char *px, int h, int w, int as_bmp) {
    return (as_bmp ? nblic::write_bmp8(path, px, h, w) : nblic::write_pgm(path, px, h, w)) ? 0 : -1;
}

}  // extern "C"

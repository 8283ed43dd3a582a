// `mathmap` command line on the HIP engine: same options, defaults and messages as the
// reference's mathmap_cmdline.c:423-895 (usage text :423-452, option table :489-522,
// -D handling :756-796, render loop :798-871), rendering through libmathmap_hip.so.
//
// PNG input/output uses a small zlib-based codec (the reference links libpng through
// rwimg/; only zlib is available here): reads 8/16-bit grey, grey+alpha, RGB, RGBA and
// palette PNGs (non-interlaced) into RGB8 -- alpha is discarded exactly like
// rwimg/rwpng.c:104-158 does -- and writes RGB8 (rwpng.c:172-251: the alpha channel of the
// RGBA render is dropped).
#include <getopt.h>
#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mmhip.h"

namespace {

int cache_size = 8;

void usage() {
    printf("Usage:\n"
           "  mathmap --version\n"
           "      print out version number\n"
           "  mathmap --help\n"
           "      print this help text\n"
           "  mathmap [option ...] [<script>] <outfile>\n"
           "      transform one or more inputs with <script> and write\n"
           "      the result to <outfile>\n"
           "  mathmap --htmldoc [<script>] <outfile>\n"
           "      outputs HTML documentation for the filters in\n"
           "      the script to <outfile>\n"
           "Options:\n"
           "  -f, --script-file=FILENAME  read script from FILENAME\n"
           "  -D<name>=<value>            define user value\n"
           "  -F, --frames=NUM            render NUM animation frames (t = frame/NUM);\n"
           "                              <outfile> may contain %%d for the frame number\n"
           "  -i, --intersampling         use intersampling\n"
           "  -o, --oversampling          use oversampling\n"
           "  -s, --size=WIDTHxHEIGHT     sets the output image size\n"
           "  -c, --cache=NUM             cache NUM input images (default %d)\n"
           "  -g, --generator=GEN         generate plug-in code with GEN\n"
           "\n"
           "Report bugs and suggestions to schani@complang.tuwien.ac.at\n",
           cache_size);
}

// ---- PNG ------------------------------------------------------------------------------------
uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

bool read_png(const char *path, std::vector<unsigned char> &rgb, int &w, int &h) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (data.size() < 8 || memcmp(data.data(), sig, 8)) return false;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    size_t p = 8;
    while (p + 12 <= data.size()) {
        uint32_t len = be32(&data[p]);
        const unsigned char *type = &data[p + 4];
        const unsigned char *body = &data[p + 8];
        if (p + 12 + len > data.size()) return false;
        if (!memcmp(type, "IHDR", 4)) {
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || interlace || (depth != 8 && depth != 16)) return false;
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (ctype == 3 && depth != 8)) return false;
    int bpp = channels * depth / 8;
    size_t stride = (size_t)w * bpp;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (int y = 0; y < h; ++y) {
        const unsigned char *src = &raw[(stride + 1) * y];
        unsigned char *dst = &img[stride * y];
        const unsigned char *up = y ? &img[stride * (y - 1)] : nullptr;
        int ft = src[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= (size_t)bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
            int x = src[i + 1], v;
            switch (ft) {
                case 0: v = x; break;
                case 1: v = x + a; break;
                case 2: v = x + b; break;
                case 3: v = x + ((a + b) >> 1); break;
                case 4: {
                    int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                    v = x + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
                    break;
                }
                default: return false;
            }
            dst[i] = (unsigned char)v;
        }
    }
    rgb.resize((size_t)w * h * 3);
    int step = depth / 8;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const unsigned char *s = &img[i * bpp];
        unsigned char *d = &rgb[i * 3];
        switch (ctype) {
            case 0: case 4: d[0] = d[1] = d[2] = s[0]; break;
            case 2: case 6: d[0] = s[0]; d[1] = s[step]; d[2] = s[2 * step]; break;
            case 3:
                if ((size_t)s[0] * 3 + 2 >= plte.size()) return false;
                d[0] = plte[s[0] * 3]; d[1] = plte[s[0] * 3 + 1]; d[2] = plte[s[0] * 3 + 2];
                break;
        }
    }
    return true;
}

void put_chunk(FILE *f, const char *type, const unsigned char *body, size_t len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8),
                            (unsigned char)len, (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2],
                            (unsigned char)type[3]};
    fwrite(hdr, 1, 8, f);
    if (len) fwrite(body, 1, len, f);
    uLong crc = crc32(0, hdr + 4, 4);
    if (len) crc = crc32(crc, body, (uInt)len);
    unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    fwrite(c, 1, 4, f);
}

bool write_png_rgb(const char *path, const unsigned char *rgba, int w, int h) {
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    fwrite(sig, 1, 8, f);
    unsigned char ihdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                              (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h,
                              8, 2, 0, 0, 0};
    put_chunk(f, "IHDR", ihdr, 13);
    std::vector<unsigned char> raw(((size_t)w * 3 + 1) * h);
    for (int y = 0; y < h; ++y) {
        unsigned char *d = &raw[((size_t)w * 3 + 1) * y];
        *d++ = 0;
        const unsigned char *s = rgba + (size_t)y * w * 4;
        for (int x = 0; x < w; ++x) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d += 3; s += 4; }
    }
    uLongf clen = compressBound(raw.size());
    std::vector<unsigned char> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) { fclose(f); return false; }
    put_chunk(f, "IDAT", comp.data(), clen);
    put_chunk(f, "IEND", nullptr, 0);
    fclose(f);
    return true;
}

struct Define { std::string name, value; };

}  // namespace

enum {
    OPT_VERSION = 256, OPT_HELP, OPT_HTMLDOC, OPT_BENCH_NO_OUTPUT, OPT_BENCH_ONLY_COMPILE,
    OPT_BENCH_NO_COMPILE_TIME_LIMIT, OPT_BENCH_NO_BACKEND, OPT_BENCH_RENDER_COUNT
};

int main(int argc, char **argv) {
    std::string script;
    bool have_script = false, htmldoc = false, bench_no_output = false, bench_no_backend = false;
    int antialiasing = 0, supersampling = 0, img_width = 0, img_height = 0, size_is_set = 0;
    int bench_render_count = 1, num_frames = 1;
    const char *generator = nullptr;
    std::vector<Define> defines;
    static struct option long_options[] = {
        {"version", no_argument, 0, OPT_VERSION}, {"help", no_argument, 0, OPT_HELP},
        {"intersampling", no_argument, 0, 'i'}, {"oversampling", no_argument, 0, 'o'},
        {"cache", required_argument, 0, 'c'}, {"generator", required_argument, 0, 'g'},
        {"size", required_argument, 0, 's'}, {"script-file", required_argument, 0, 'f'},
        {"htmldoc", no_argument, 0, OPT_HTMLDOC}, {"bench-no-output", no_argument, 0, OPT_BENCH_NO_OUTPUT},
        {"bench-only-compile", no_argument, 0, OPT_BENCH_ONLY_COMPILE},
        {"bench-no-compile-time-limit", no_argument, 0, OPT_BENCH_NO_COMPILE_TIME_LIMIT},
        {"bench-no-backend", no_argument, 0, OPT_BENCH_NO_BACKEND},
        {"bench-render-count", required_argument, 0, OPT_BENCH_RENDER_COUNT},
        {"frames", required_argument, 0, 'F'}, {0, 0, 0, 0}};
    for (;;) {
        int idx;
        int option = getopt_long(argc, argv, "f:ioF:D:c:g:s:", long_options, &idx);
        if (option == -1) break;
        switch (option) {
            case OPT_VERSION:
                printf("MathMap (HIP backend) %s\n", mmhip_version());
                return 0;
            case OPT_HELP: usage(); return 0;
            case OPT_HTMLDOC: htmldoc = true; break;
            case 'f': {
                FILE *f = fopen(optarg, "rb");
                if (!f) { fprintf(stderr, "Error: The script file `%s' could not be read.\n", optarg); return 1; }
                char buf[4096];
                size_t n;
                while ((n = fread(buf, 1, sizeof buf, f)) > 0) script.append(buf, n);
                fclose(f);
                have_script = true;
                break;
            }
            case 'i': antialiasing = 1; break;
            case 'o': supersampling = 1; break;
            case 'c': cache_size = atoi(optarg); break;
            case 'D': {
                const char *eq = strchr(optarg, '=');
                if (!eq) { fprintf(stderr, "Error: No equal sign in -D option.\n"); return 1; }
                defines.push_back({std::string(optarg, eq - optarg), std::string(eq + 1)});
                break;
            }
            case 'g': generator = optarg; break;
            case 's':
                if (sscanf(optarg, "%dx%d", &img_width, &img_height) != 2 || img_width <= 0 || img_height <= 0) {
                    fprintf(stderr, "Error: Invalid image size.  Syntax is <width>x<height>.  Example: 1024x768.\n");
                    return 1;
                }
                size_is_set = 1;
                break;
            case 'F': num_frames = atoi(optarg); if (num_frames < 1) num_frames = 1; break;
            case OPT_BENCH_RENDER_COUNT: bench_render_count = atoi(optarg); break;
            case OPT_BENCH_ONLY_COMPILE: bench_render_count = 0; break;
            case OPT_BENCH_NO_OUTPUT: bench_no_output = true; break;
            case OPT_BENCH_NO_COMPILE_TIME_LIMIT: break;
            case OPT_BENCH_NO_BACKEND: bench_no_backend = true; break;
            default: usage(); return 1;
        }
    }
    const char *output_filename;
    if (have_script) {
        if (argc - optind != 1) { usage(); return 1; }
        output_filename = argv[optind];
    } else {
        if (argc - optind != 2) { usage(); return 1; }
        script = argv[optind];
        output_filename = argv[optind + 1];
    }
    if (htmldoc) { fprintf(stderr, "Error: --htmldoc is not supported by the HIP command line.\n"); return 1; }
    if (generator) { fprintf(stderr, "Unknown generator `%s'\n", generator); return 1; }

    mmhip_options opts;
    mmhip_default_options(&opts);
    opts.intersample = antialiasing;
    opts.supersampling = supersampling;
    // an extension next to the reference's options: a script whose first character is '{' (which no .mm text can
    // start with) is a compiled filter in the IR dump form of mmhip_filter_ir_json_raw, e.g. the output of another
    // front-end; everything after compilation is the same
    size_t first = script.find_first_not_of(" \t\r\n");
    mmhip_filter *flt = (first != std::string::npos && script[first] == '{') ? mmhip_compile_ir_json(script.c_str(), &opts)
                                                                             : mmhip_compile(script.c_str(), &opts);
    if (bench_no_backend) return 0;
    if (!flt) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
    if (bench_render_count == 0) return mmhip_filter_jit(flt, 0) < 0 ? 1 : 0;

    auto lookup = [&](const char *name) -> const Define * {
        for (const Define &d : defines) if (d.name == name) return &d;
        return nullptr;
    };
    int nuv = mmhip_filter_num_uservals(flt);
    std::vector<unsigned char> pixels;
    if (!size_is_set)
        for (int i = 0; i < nuv; ++i) {
            mmhip_userval_info info;
            mmhip_filter_userval_info(flt, i, &info);
            if (info.kind != MMHIP_UV_IMAGE) continue;
            const Define *d = lookup(info.name);
            if (!d) { fprintf(stderr, "Error: No value defined for input image `%s'.\n", info.name); return 1; }
            if (!read_png(d->value.c_str(), pixels, img_width, img_height)) {
                fprintf(stderr, "Error: Could not read input image `%s'.\n", d->value.c_str());
                return 1;
            }
            size_is_set = 1;
            break;
        }
    if (!size_is_set) { fprintf(stderr, "Error: Image size not set and no input images given.\n"); return 1; }

    mmhip_invocation *inv = mmhip_invoke(flt, img_width, img_height);
    if (!inv) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
    for (int i = 0; i < nuv; ++i) {
        mmhip_userval_info info;
        mmhip_filter_userval_info(flt, i, &info);
        const Define *d = lookup(info.name);
        if (!d) {
            if (info.kind == MMHIP_UV_IMAGE) { fprintf(stderr, "Error: No value defined for input image `%s'.\n", info.name); return 1; }
            continue;
        }
        switch (info.kind) {
            case MMHIP_UV_INT: mmhip_set_int(inv, i, atoi(d->value.c_str())); break;
            case MMHIP_UV_FLOAT: mmhip_set_float(inv, i, (float)strtod(d->value.c_str(), nullptr)); break;
            case MMHIP_UV_BOOL: mmhip_set_bool(inv, i, (int)(float)atoi(d->value.c_str())); break;
            case MMHIP_UV_IMAGE: {
                int w, h;
                if (!read_png(d->value.c_str(), pixels, w, h)) {
                    fprintf(stderr, "Error: Could not read input image `%s'.\n", d->value.c_str());
                    return 1;
                }
                if (mmhip_set_image_host(inv, i, pixels.data(), w, h, 3) != 0) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
                break;
            }
            default:
                fprintf(stderr, "Error: Can only define user values for types int, float, bool and image.\n");
                return 1;
        }
    }

    std::vector<unsigned char> output((size_t)img_width * img_height * 4);
    void *dev = mmhip_device_alloc(output.size());
    if (!dev) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
    for (int render_num = 0; render_num < bench_render_count; ++render_num) {
        for (int frame = 0; frame < num_frames; ++frame) {
            float t = (float)frame / (float)num_frames;       // mathmap_cmdline.c:835
            int rc = supersampling
                         ? mmhip_render_supersampled(inv, frame, t, 0, 0, img_width, img_height, dev, img_width * 4, 4, nullptr)
                         : mmhip_render(inv, frame, t, 0, 0, img_width, img_height, 0, img_height, dev, img_width * 4, 4, 0, nullptr);
            if (rc != 0 || mmhip_sync(inv) != 0) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
            if (bench_no_output) continue;
            if (mmhip_copy_to_host(output.data(), dev, output.size()) != 0) { fprintf(stderr, "Error: %s\n", mmhip_last_error()); return 1; }
            char name[4096];
            if (num_frames > 1 && strstr(output_filename, "%")) snprintf(name, sizeof name, output_filename, frame);
            else snprintf(name, sizeof name, "%s", output_filename);
            if (!write_png_rgb(name, output.data(), img_width, img_height)) {
                fprintf(stderr, "Error: Cannot open file `%s' for writing: %s\n", name, strerror(errno));
                return 1;
            }
        }
    }
    mmhip_device_free(dev);
    mmhip_invocation_free(inv);
    mmhip_filter_free(flt);
    return 0;
}

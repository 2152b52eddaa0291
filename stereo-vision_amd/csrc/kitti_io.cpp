// KITTI raw input in front of the hot path (include/svh_kitti.h): the rectified
// camera calibration, the gray PNG frames and the timestamp lists, read the way stereomapper's
// playback thread does (readfromfilesthread.cpp:25-112) but with zlib instead of OpenCV/Qt.
// Host only: no HIP in this file.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <vector>

#include "../../include/svh.h"
#include "../../include/svh_kitti.h"

namespace {

// ---------------------------------------------------------------------------
// calibration text: "name: v v v ..." lines, tokens separated by blank, tab, comma or semicolon
// (calibiokitti.cpp:110-138)
// ---------------------------------------------------------------------------
std::vector<std::string> tokens_of(const char* line) {
    std::vector<std::string> out;
    std::string cur;
    for (const char* c = line;; c++) {
        const bool sep = *c == ' ' || *c == '\t' || *c == ',' || *c == ';' || *c == '\n' || *c == '\r' || *c == '\0';
        if (!sep) {
            cur.push_back(*c);
        } else if (!cur.empty()) {
            out.push_back(cur);
            cur.clear();
        }
        if (*c == '\0' || *c == '\n') break;
    }
    return out;
}

// the first line whose leading token is `name`; every lookup scans from the top of the file, so
// the order of the entries does not matter (calibiokitti.cpp:176-224)
bool find_entry(const std::vector<std::string>& lines, const std::string& name, std::vector<std::string>& tok) {
    for (const std::string& l : lines) {
        tok = tokens_of(l.c_str());
        if (!tok.empty() && tok[0] == name) return true;
    }
    return false;
}

bool read_matrix(const std::vector<std::string>& lines, const std::string& name, size_t count, double* out) {
    std::vector<std::string> tok;
    if (!find_entry(lines, name, tok)) return false;
    if (tok.size() - 1 != count) {
        printf("ERROR Number of elements in %s %zu!=%zu\n", name.c_str(), tok.size() - 1, count);
        return false;
    }
    // the reference extracts every value into a float before widening it (calibiokitti.cpp:201-212)
    for (size_t i = 0; i < count; i++) out[i] = (double)strtof(tok[i + 1].c_str(), nullptr);
    return true;
}

bool read_lines(const char* path, std::vector<std::string>& lines) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    std::vector<char> buf(20000);
    while (fgets(buf.data(), (int)buf.size(), f)) lines.emplace_back(buf.data());
    fclose(f);
    return true;
}

// ---------------------------------------------------------------------------
// PNG (RFC 2083): signature, IHDR, concatenated IDAT -> zlib stream -> per-row filters
// ---------------------------------------------------------------------------
uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

struct PngInfo {
    int32_t w = 0, h = 0;
    int depth = 0, colour = 0, channels = 0;
};

int32_t png_decode(const std::vector<uint8_t>& file, PngInfo& info, std::vector<uint8_t>& raw) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return SVH_ERR_BAD_ARG;
    std::vector<uint8_t> idat;
    bool have_hdr = false, ended = false;
    for (size_t at = 8; at + 12 <= file.size() && !ended;) {
        const uint32_t len = be32(&file[at]);
        const uint8_t* type = &file[at + 4];
        if (at + 12 + (size_t)len > file.size()) return SVH_ERR_BAD_ARG;
        const uint8_t* body = &file[at + 8];
        if (be32(body + len) != (uint32_t)crc32(crc32(0, Z_NULL, 0), type, len + 4)) return SVH_ERR_BAD_ARG;
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return SVH_ERR_BAD_ARG;
            info.w = (int32_t)be32(body);
            info.h = (int32_t)be32(body + 4);
            info.depth = body[8];
            info.colour = body[9];
            if (body[10] != 0 || body[11] != 0) return SVH_ERR_BAD_ARG;   // compression / filter method
            if (body[12] != 0) return SVH_ERR_UNSUPPORTED;                // Adam7
            have_hdr = true;
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            ended = true;
        }
        at += 12 + (size_t)len;
    }
    if (!have_hdr || !ended || info.w <= 0 || info.h <= 0) return SVH_ERR_BAD_ARG;
    switch (info.colour) {
        case 0: info.channels = 1; break;
        case 2: info.channels = 3; break;
        case 4: info.channels = 2; break;
        case 6: info.channels = 4; break;
        default: return SVH_ERR_UNSUPPORTED;   // palette
    }
    if (info.depth != 8 && info.depth != 16) return SVH_ERR_UNSUPPORTED;
    // a corrupt or crafted header must not size the buffers: 2^28 pixels is far beyond any camera frame
    if ((uint64_t)info.w * (uint64_t)info.h > (1ull << 28)) return SVH_ERR_BAD_ARG;
    const size_t bpp = (size_t)info.channels * info.depth / 8, stride = bpp * info.w;
    std::vector<uint8_t> flat((stride + 1) * info.h);
    uLongf got = (uLongf)flat.size();
    if (uncompress(flat.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != flat.size())
        return SVH_ERR_BAD_ARG;
    raw.assign(stride * info.h, 0);
    for (int32_t y = 0; y < info.h; y++) {
        const uint8_t* in = &flat[(stride + 1) * y];
        uint8_t* cur = &raw[stride * y];
        const uint8_t* up = y ? cur - stride : nullptr;
        const int filter = in[0];
        if (filter > 4) return SVH_ERR_BAD_ARG;
        in++;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            if (filter == 1) pred = a;
            else if (filter == 2) pred = b;
            else if (filter == 3) pred = (a + b) >> 1;
            else if (filter == 4) pred = paeth(a, b, c);
            cur[i] = (uint8_t)(in[i] + pred);
        }
    }
    return SVH_OK;
}

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// "2011-09-26 13:02:25.964389445" -> time of day; the fields sit at fixed columns
// (stereoimageiokitti.cpp:100-105)
bool parse_stamp(const std::string& line, int64_t* sec, int64_t* usec) {
    if (line.size() < 29) return false;
    auto num = [&](size_t at, size_t n) { return atoll(line.substr(at, n).c_str()); };
    *sec = num(11, 2) * 3600 + num(14, 2) * 60 + num(17, 2);
    *usec = num(20, 9) / 1000;
    return true;
}

}  // namespace

struct svh_kitti_seq {
    std::string dir[2];
    std::vector<std::string> stamps[2];
    int32_t next = 0;
};

extern "C" {

int32_t svh_kitti_read_cam_to_cam(const char* path, svh_kitti_calib* out) {
    try {
    if (!path || !out) return SVH_ERR_BAD_ARG;
    std::vector<std::string> lines;
    if (!read_lines(path, lines)) return SVH_ERR_BAD_ARG;
    memset(out, 0, sizeof(*out));
    std::vector<std::string> tok;
    if (find_entry(lines, "calib_time:", tok)) {
        std::string t;
        for (size_t i = 1; i < tok.size(); i++) t += (i > 1 ? " " : "") + tok[i];
        snprintf(out->calib_time, sizeof(out->calib_time), "%s", t.c_str());
    }
    bool ok = read_matrix(lines, "corner_dist:", 1, &out->corner_dist);
    for (int i = 0; i < SVH_KITTI_CAMERAS; i++) {
        const std::string n = "_0" + std::to_string(i) + ":";
        ok = read_matrix(lines, "S" + n, 2, out->S[i]) && ok;
        ok = read_matrix(lines, "K" + n, 9, out->K[i]) && ok;
        ok = read_matrix(lines, "D" + n, 5, out->D[i]) && ok;
        ok = read_matrix(lines, "R" + n, 9, out->R[i]) && ok;
        ok = read_matrix(lines, "T" + n, 3, out->T[i]) && ok;
        ok = read_matrix(lines, "S_rect" + n, 2, out->S_rect[i]) && ok;
        ok = read_matrix(lines, "R_rect" + n, 9, out->R_rect[i]) && ok;
        ok = read_matrix(lines, "P_rect" + n, 12, out->P_rect[i]) && ok;
    }
    if (!ok) return SVH_ERR_BAD_ARG;
    // stereothread.cpp:444-447 (members are float there; the quotient is formed in double first)
    out->f = out->P_rect[0][0];
    out->cu = out->P_rect[0][2];
    out->cv = out->P_rect[0][6];
    out->base = -out->P_rect[1][3] / out->P_rect[1][0];
    return SVH_OK;
    } catch (...) {   // std::bad_alloc etc. must not cross the C boundary
        return SVH_ERR_BAD_ARG;
    }
}

int32_t svh_png_read_gray(const char* path, uint8_t* buf, size_t cap, int32_t* width, int32_t* height) {
    try {
    if (!path || !width || !height) return SVH_ERR_BAD_ARG;
    std::vector<uint8_t> file, raw;
    if (!read_file(path, file)) return SVH_ERR_BAD_ARG;
    PngInfo info;
    const int32_t rc = png_decode(file, info, raw);
    if (rc != SVH_OK) return rc;
    *width = info.w;
    *height = info.h;
    if (!buf) return SVH_OK;
    const size_t n = (size_t)info.w * info.h;
    if (cap < n) return SVH_ERR_BAD_ARG;
    const size_t bps = info.depth / 8, px = bps * info.channels;   // big-endian samples: high byte first
    for (size_t i = 0; i < n; i++) {
        const uint8_t* p = &raw[i * px];
        if (info.channels <= 2) {
            buf[i] = p[0];
        } else {
            const uint32_t r = p[0], g = p[bps], b = p[2 * bps];
            buf[i] = (uint8_t)((r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14);
        }
    }
    return SVH_OK;
    } catch (...) {   // std::bad_alloc etc. must not cross the C boundary
        return SVH_ERR_BAD_ARG;
    }
}

svh_kitti_seq* svh_kitti_seq_open(const char* drive_dir) {
    try {
    if (!drive_dir) return nullptr;
    svh_kitti_seq* s = new svh_kitti_seq();
    for (int k = 0; k < 2; k++) {
        const std::string cam = std::string(drive_dir) + "/image_0" + std::to_string(k);
        s->dir[k] = cam + "/data";
        std::vector<std::string> lines;
        if (!read_lines((cam + "/timestamps.txt").c_str(), lines)) {
            printf("ERROR: cannot open timestamp text file\n");
            delete s;
            return nullptr;
        }
        // a frame is a newline-terminated line (the reference counts '\n')
        for (const std::string& l : lines)
            if (!l.empty() && l.back() == '\n') s->stamps[k].push_back(l);
    }
    if (s->stamps[0].size() != s->stamps[1].size()) {
        printf("ERROR: timestamp lines counts are not consistent\n");
        delete s;
        return nullptr;
    }
    return s;
    } catch (...) {   // std::bad_alloc etc. must not cross the C boundary
        return nullptr;
    }
}

void svh_kitti_seq_close(svh_kitti_seq* s) { delete s; }

int32_t svh_kitti_seq_count(const svh_kitti_seq* s) { return s ? (int32_t)s->stamps[0].size() : 0; }

int32_t svh_kitti_seq_seek(svh_kitti_seq* s, int32_t frame) {
    if (!s || frame < 0 || frame > (int32_t)s->stamps[0].size()) return SVH_ERR_BAD_ARG;
    s->next = frame;
    return SVH_OK;
}

int32_t svh_kitti_seq_next(svh_kitti_seq* s, uint8_t* I1, uint8_t* I2, size_t cap, int32_t* dims, int64_t* tv) {
    try {
    if (!s || !I1 || !I2 || !dims) return SVH_ERR_BAD_ARG;
    if (s->next >= (int32_t)s->stamps[0].size()) return 1;
    const int32_t idx = s->next++;
    int32_t w[2] = {0, 0}, h[2] = {0, 0};
    uint8_t* dst[2] = {I1, I2};
    for (int k = 0; k < 2; k++) {
        int64_t sec = 0, usec = 0;
        if (!parse_stamp(s->stamps[k][idx], &sec, &usec)) return SVH_ERR_BAD_ARG;
        if (tv) {
            tv[2 * k] = sec;
            tv[2 * k + 1] = usec;
        }
        char name[32];
        snprintf(name, sizeof(name), "/%010d.png", idx);
        const int32_t rc = svh_png_read_gray((s->dir[k] + name).c_str(), dst[k], cap, &w[k], &h[k]);
        if (rc != SVH_OK) return rc;
    }
    if (w[0] != w[1] || h[0] != h[1]) return SVH_ERR_BAD_ARG;
    dims[0] = w[0];
    dims[1] = h[0];
    dims[2] = w[0];
    return SVH_OK;
    } catch (...) {   // std::bad_alloc etc. must not cross the C boundary
        return SVH_ERR_BAD_ARG;
    }
}

}  // extern "C"

// Boundary test: a Qt-free replica of the reference call sites
//   libelas/src/main.cpp:55-64          (demo: postprocess_only_left = false)
//   stereomapper/stereothread.cpp:76-114 (per-frame Elas, ROBOTICS,
//                                         postprocess_only_left, adaptive mean,
//                                         support_texture = 30, optional subsampling)
// compiled against include/elas.h exactly as those files include "elas.h".
//
//   elas_dropin <left.pgm> <right.pgm> <mode: demo|mapper> <out_D1.f32> <out_D2.f32>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "elas.h"

static bool read_pgm(const char* path, std::vector<uint8_t>& img, int32_t& w, int32_t& h) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    if (fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) != 4 || strcmp(magic, "P5") != 0) {
        fclose(f);
        return false;
    }
    fgetc(f);
    img.resize((size_t)w * h);
    bool ok = fread(img.data(), 1, img.size(), f) == img.size();
    fclose(f);
    return ok;
}

static void write_raw(const char* path, const float* d, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) return;
    fwrite(d, sizeof(float), n, f);
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc != 6) {
        fprintf(stderr, "usage: %s left.pgm right.pgm demo|mapper D1.f32 D2.f32\n", argv[0]);
        return 2;
    }
    std::vector<uint8_t> I1, I2;
    int32_t width = 0, height = 0, w2 = 0, h2 = 0;
    if (!read_pgm(argv[1], I1, width, height) || !read_pgm(argv[2], I2, w2, h2) || w2 != width ||
        h2 != height) {
        fprintf(stderr, "cannot read the input pair\n");
        return 2;
    }
    const int32_t dims[3] = {width, height, width};
    float* D1_data = (float*)malloc(width * height * sizeof(float));
    float* D2_data = (float*)malloc(width * height * sizeof(float));
    for (int32_t i = 0; i < width * height; i++) D1_data[i] = D2_data[i] = -7.f;

    if (!strcmp(argv[3], "demo")) {
        // libelas/src/main.cpp:61-64
        Elas::parameters param;
        param.postprocess_only_left = false;
        Elas elas(param);
        elas.process(I1.data(), I2.data(), D1_data, D2_data, dims);
    } else {
        // stereomapper/stereothread.cpp:76-80, 113-114
        Elas::parameters param(Elas::ROBOTICS);
        param.postprocess_only_left = true;
        param.filter_adaptive_mean = true;
        param.support_texture = 30;
        param.subsampling = false;
        Elas elas(param);
        elas.process(I1.data(), I2.data(), D1_data, D2_data, dims);
    }
    write_raw(argv[4], D1_data, (size_t)width * height);
    write_raw(argv[5], D2_data, (size_t)width * height);
    free(D1_data);
    free(D2_data);
    return 0;
}

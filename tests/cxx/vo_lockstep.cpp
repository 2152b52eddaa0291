// K sequences in one process through the C++ classes of include/viso_stereo.h: the frame loop of
// libviso2/src/demo.cpp:86-131 (read the images of frame i, viso.process(...), getMotion()) for K
// VisualOdometryStereo objects -- once as K process() calls per frame, once as ONE processBatch() per frame, once as
// the pipelined loop (prefetchBatch / processNextBatch).  With equal srand() the three must agree bit for bit.
//
//   vo_lockstep I1p.pgm I2p.pgm I1c.pgm I2c.pgm [K] [frames]      prints "vo_lockstep: OK ..." or the mismatch
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "viso_stereo.h"

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
    const bool ok = fread(img.data(), 1, img.size(), f) == img.size();
    fclose(f);
    return ok;
}

struct Log {   // what the caller of a frame sees, per object
    std::vector<int> ok;
    std::vector<double> motion;        // 16 per (frame, object)
    std::vector<int32_t> inliers;      // count, then the indices
};

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s I1p I2p I1c I2c [K] [frames]\n", argv[0]);
        return 2;
    }
    const int K = argc > 5 ? atoi(argv[5]) : 6, frames = argc > 6 ? atoi(argv[6]) : 7;
    std::vector<uint8_t> im[4];
    int32_t w = 0, h = 0;
    for (int k = 0; k < 4; k++)
        if (!read_pgm(argv[1 + k], im[k], w, h)) {
            fprintf(stderr, "cannot read %s\n", argv[1 + k]);
            return 2;
        }
    // sequence k: the quad shifted by 3k columns (cyclically), frames alternate previous / current pair
    std::vector<std::vector<uint8_t> > seq((size_t)K * 4);
    for (int k = 0; k < K; k++)
        for (int q = 0; q < 4; q++) {
            std::vector<uint8_t>& d = seq[(size_t)k * 4 + q];
            d.resize(im[q].size());
            for (int32_t v = 0; v < h; v++)
                for (int32_t u = 0; u < w; u++) d[(size_t)v * w + (u + 3 * k) % w] = im[q][(size_t)v * w + u];
        }
    VisualOdometryStereo::parameters param;   // calibration of libviso2/src/demo.cpp:54-58
    param.calib.f = 645.24;
    param.calib.cu = 635.96;
    param.calib.cv = 194.13;
    param.base = 0.5707;
    int32_t dims[3] = {w, h, w};

    Log log[3];
    for (int mode = 0; mode < 3; mode++) {
        std::vector<VisualOdometryStereo*> vos(K);
        for (int k = 0; k < K; k++) vos[k] = new VisualOdometryStereo(param);
        srand(4711);   // (the constructors called srand(0), viso.cpp:36)
        std::vector<uint8_t*> I1(K), I2(K), N1(K), N2(K);
        std::vector<int32_t> ok(K);
        auto frame_ptrs = [&](int i, std::vector<uint8_t*>& a, std::vector<uint8_t*>& b) {
            for (int k = 0; k < K; k++) {
                a[k] = seq[(size_t)k * 4 + (i % 2 ? 2 : 0)].data();
                b[k] = seq[(size_t)k * 4 + (i % 2 ? 3 : 1)].data();
            }
        };
        if (mode == 2) {
            frame_ptrs(0, I1, I2);
            if (VisualOdometryStereo::prefetchBatch(vos.data(), K, I1.data(), I2.data(), dims) < 0) return 3;
        }
        for (int i = 0; i < frames; i++) {
            frame_ptrs(i, I1, I2);
            if (mode == 0) {
                for (int k = 0; k < K; k++) ok[k] = vos[k]->process(I1[k], I2[k], dims, false) ? 1 : 0;
            } else if (mode == 1) {
                if (VisualOdometryStereo::processBatch(vos.data(), K, I1.data(), I2.data(), dims, false, ok.data()) < 0) return 3;
            } else {
                const bool more = i + 1 < frames;
                if (more) frame_ptrs(i + 1, N1, N2);
                if (VisualOdometryStereo::processNextBatch(vos.data(), K, more ? N1.data() : 0, more ? N2.data() : 0, dims,
                                                           false, ok.data()) < 0)
                    return 3;
            }
            for (int k = 0; k < K; k++) {
                log[mode].ok.push_back(ok[k]);
                Matrix T = vos[k]->getDeltaMotion();
                for (int r = 0; r < 4; r++)
                    for (int c = 0; c < 4; c++) log[mode].motion.push_back(T._val[r][c]);
                std::vector<int32_t> inl = vos[k]->getInlierIndices();
                log[mode].inliers.push_back((int32_t)inl.size());
                log[mode].inliers.insert(log[mode].inliers.end(), inl.begin(), inl.end());
            }
        }
        for (int k = 0; k < K; k++) delete vos[k];
    }
    int good = 0;
    for (size_t i = 0; i < log[0].ok.size(); i++) good += log[0].ok[i];
    for (int mode = 1; mode < 3; mode++) {
        const char* name = mode == 1 ? "processBatch" : "processNextBatch";
        if (log[mode].ok != log[0].ok) { printf("vo_lockstep: %s: return values differ\n", name); return 1; }
        if (log[mode].motion.size() != log[0].motion.size() ||
            memcmp(log[mode].motion.data(), log[0].motion.data(), log[0].motion.size() * sizeof(double)) != 0) {
            printf("vo_lockstep: %s: motions differ\n", name);
            return 1;
        }
        if (log[mode].inliers != log[0].inliers) { printf("vo_lockstep: %s: inlier sets differ\n", name); return 1; }
    }
    printf("vo_lockstep: OK %d objects x %d frames, %d motion updates, three loops bit-identical\n", K, frames, good);
    return good >= K * (frames - 2) ? 0 : 1;
}

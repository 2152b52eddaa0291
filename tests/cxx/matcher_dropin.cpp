// Boundary test: a replica of VisualOdometryStereo::process's use of the Matcher
// (libviso2/src/viso_stereo.cpp:41-68, viso.cpp:33-36) compiled against
// include/matcher.h + include/matrix.h exactly as the reference includes them.
//
//   matcher_dropin I1p.pgm I2p.pgm I1c.pgm I2c.pgm out_matches.bin [predict]
//
// Writes the p_match list of the second frame (before bucketing) as raw structs,
// then "bucketed <n>" on stdout after bucketFeatures(2,50,50) with srand(0).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "matcher.h"

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

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s I1p I2p I1c I2c out.bin [predict]\n", argv[0]);
        return 2;
    }
    std::vector<uint8_t> im[4];
    int32_t w = 0, h = 0;
    for (int k = 0; k < 4; k++)
        if (!read_pgm(argv[1 + k], im[k], w, h)) {
            fprintf(stderr, "cannot read %s\n", argv[1 + k]);
            return 2;
        }
    const bool predict = argc > 6;

    // viso.cpp:33-36: one long-lived matcher per VO object
    Matcher::parameters param;
    Matcher* _matcher = new Matcher(param);
    _matcher->setIntrinsics(645.24, 635.96, 194.13, 0.5707);

    int32_t dims[3] = {w, h, w};
    // frame 0 (viso_stereo.cpp:44): nothing to match yet
    _matcher->pushBack(im[0].data(), im[1].data(), dims, false);
    _matcher->matchFeatures(2);
    std::vector<Matcher::p_match> _p_matched = _matcher->getMatches();
    if (!_p_matched.empty()) return 3;

    // frame 1 (viso_stereo.cpp:44-66)
    _matcher->pushBack(im[2].data(), im[3].data(), dims, false);
    Matrix _Tr_delta = Matrix::eye(4);
    _Tr_delta._val[2][3] = -0.75;
    if (predict) _matcher->matchFeatures(2, &_Tr_delta);
    else         _matcher->matchFeatures(2);
    _p_matched = _matcher->getMatches();
    FILE* f = fopen(argv[5], "wb");
    if (!f) return 2;
    fwrite(_p_matched.data(), sizeof(Matcher::p_match), _p_matched.size(), f);
    fclose(f);
    // viso.cpp:36 seeds libc's rand() once at start-up; here it is re-seeded right before the
    // shuffle so that the comparison does not depend on other rand() users in the process
    srand(0);
    _matcher->bucketFeatures(2, 50, 50);
    _p_matched = _matcher->getMatches();
    printf("matches %zu\n", _p_matched.size());
    std::vector<int32_t> inliers;
    for (int32_t i = 0; i < (int32_t)_p_matched.size(); i += 3) inliers.push_back(i);
    printf("gain %.6f\n", _matcher->getGain(inliers));
    delete _matcher;
    return 0;
}

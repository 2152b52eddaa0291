// Boundary test: a replica of stereomapper's VisualOdometryThread (stereomapper/
// visualodometrythread.cpp:19-49 constructor, :92-131 run) compiled against
// include/viso_stereo.h exactly as the reference includes it.
//
//   vo_dropin I1p.pgm I2p.pgm I1c.pgm I2c.pgm out.bin
//
// out.bin: 16 doubles H_Delta (row major), 3 doubles roll/pitch/yaw, 1 double velocity,
// 1 float gain, int32 n_matches, int32 n_inliers, then the inlier indices (int32).
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
    bool ok = fread(img.data(), 1, img.size(), f) == img.size();
    fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s I1p I2p I1c I2c out.bin\n", argv[0]);
        return 2;
    }
    std::vector<uint8_t> im[4];
    int32_t w = 0, h = 0;
    for (int k = 0; k < 4; k++)
        if (!read_pgm(argv[1 + k], im[k], w, h)) {
            fprintf(stderr, "cannot read %s\n", argv[1 + k]);
            return 2;
        }

    // visualodometrythread.cpp:19-49 (calibration of libviso2/src/demo.cpp:54-58)
    VisualOdometryStereo::parameters visualOdomStereoParam;
    visualOdomStereoParam.calib.f = 645.24;
    visualOdomStereoParam.calib.cu = 635.96;
    visualOdomStereoParam.calib.cv = 194.13;
    visualOdomStereoParam.base = 0.5707;
    VisualOdometryStereo* _visualOdomStereo = new VisualOdometryStereo(visualOdomStereoParam);

    int32_t dim[3] = {w, h, w};
    // first frame: nothing to match against, process() reports failure
    if (_visualOdomStereo->process(im[0].data(), im[1].data(), dim, false)) return 3;

    // visualodometrythread.cpp:104-131
    const bool ok = _visualOdomStereo->process(im[2].data(), im[3].data(), dim, false);
    Matrix H_Delta_inv = Matrix::eye(4);
    Matrix H_Delta = _visualOdomStereo->getDeltaMotion();
    double roll, pitch, yaw, vel;
    _visualOdomStereo->calculateRollPitchYawFromTransformation(roll, pitch, yaw);
    _visualOdomStereo->calculateVelocityFromTransformation(vel);
    std::vector<int32_t> inliers = _visualOdomStereo->getInlierIndices();
    std::vector<bool> mask;
    for (int32_t i = 0; i < (int32_t)_visualOdomStereo->getMatches().size(); i++) mask.push_back(false);
    for (std::vector<int32_t>::iterator it = inliers.begin(); it != inliers.end(); it++) mask[*it] = true;
    float gain = _visualOdomStereo->getGain(inliers);
    if (!H_Delta_inv.solve(H_Delta)) return 4;

    FILE* f = fopen(argv[5], "wb");
    if (!f) return 2;
    Matrix H = _visualOdomStereo->getDeltaMotion();
    for (int i = 0; i < 4; i++) fwrite(H._val[i], sizeof(double), 4, f);
    double rpyv[4] = {roll, pitch, yaw, vel};
    fwrite(rpyv, sizeof(double), 4, f);
    fwrite(&gain, sizeof(float), 1, f);
    int32_t n[2] = {_visualOdomStereo->getNumberOfMatches(), _visualOdomStereo->getNumberOfInliers()};
    fwrite(n, sizeof(int32_t), 2, f);
    if (!inliers.empty()) fwrite(inliers.data(), sizeof(int32_t), inliers.size(), f);
    fclose(f);
    printf("ok %d matches %d inliers %d\n", ok ? 1 : 0, n[0], n[1]);
    delete _visualOdomStereo;
    return ok ? 0 : 5;
}

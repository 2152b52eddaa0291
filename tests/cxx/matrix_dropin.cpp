// Call sites of the reference's applications that use Matrix next to the hot path, replayed
// on a header named "matrix.h":
//   stereomapper/stereothread.cpp:303-307   Matrix::inv(H), _K * H.getMat(0,0,2,3)
//   stereomapper/maindialog.cpp:396-406     setDiag / setMat / reshape / eye(4) / inv chain
//   stereomapper/view3d.cpp:93              rotMatY(-ry) * rotMatX(-rx)
//   stereomapper/planeestimation.cpp:98-117 operator/ by l2norm, cross, eye(), setMat
// plus the remaining public members.  Every element is printed as a hex double, so two builds
// can be compared bit for bit: this file is compiled once against include/matrix.h (the
// product's header-only class) and once against the reference's matrix.h + matrix.cpp
// (oracle/Makefile, build container only); tests/golden/matrix_dropin.txt is the latter's output.
#include <math.h>
#include <stdio.h>

#include <iostream>
#include <vector>

#include "matrix.h"

static void dump(const char* name, Matrix M) {
    printf("%s %dx%d:", name, M._m, M._n);
    for (int32_t i = 0; i < M._m; i++)
        for (int32_t j = 0; j < M._n; j++) printf(" %a", M._val[i][j]);
    printf("\n");
}

int main() {
    // a camera pose: rotation about all three axes and a translation
    Matrix H = Matrix::eye(4);
    H.setMat(Matrix::rotMatZ(0.03) * Matrix::rotMatY(-0.41) * Matrix::rotMatX(0.17), 0, 0);
    H._val[0][3] = 0.7; H._val[1][3] = -0.12; H._val[2][3] = 5.3;
    FLOAT Kd[9] = {721.5377, 0, 609.5593, 0, 721.5377, 172.854, 0, 0, 1};
    Matrix K(3, 3, Kd);
    // stereothread.cpp:303-307
    Matrix Hi = Matrix::inv(H);
    Matrix P = K * Hi.getMat(0, 0, 2, 3);
    dump("H", H); dump("inv(H)", Hi); dump("P", P);
    // maindialog.cpp:396-406
    FLOAT Rd[9] = {7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04, -9.998902e-01,
                   9.998621e-01, 7.523790e-03, 1.480755e-02};
    FLOAT Td[3] = {-4.069766e-03, -7.631618e-02, -2.717806e-01};
    Matrix R(3, 3, Rd), T(1, 3, Td);
    Matrix velo_to_cam(4, 4);
    velo_to_cam.setDiag(1.0F);
    velo_to_cam.setMat(R, 0, 0);
    velo_to_cam.setMat(Matrix::reshape(T, 3, 1), 0, 3);
    Matrix cam_to_velo = Matrix::eye(4);
    cam_to_velo = Matrix::inv(velo_to_cam);
    dump("velo_to_cam", velo_to_cam); dump("cam_to_velo", cam_to_velo); dump("product", velo_to_cam * cam_to_velo);
    // view3d.cpp:93
    float rx = 200.0f * M_PI / 180.0, ry = -33.5f * M_PI / 180.0;
    Matrix Rv = Matrix::rotMatY(-ry) * Matrix::rotMatX(-rx);
    Matrix v(3, 1);
    v._val[0][0] = 0.5; v._val[1][0] = -1.25; v._val[2][0] = 3.0;
    dump("Rv", Rv); dump("Rv*v", Rv * v);
    // planeestimation.cpp:98-117
    Matrix e(3, 1);
    e._val[0][0] = 0.02; e._val[1][0] = -0.98; e._val[2][0] = 0.11;
    Matrix r2 = e / e.l2norm();
    Matrix r1(3, 1);
    r1._val[0][0] = +sqrt(r2._val[1][0] * r2._val[1][0] / (r2._val[0][0] * r2._val[0][0] + r2._val[1][0] * r2._val[1][0]));
    r1._val[1][0] = -r1._val[0][0] * r2._val[0][0] / r2._val[1][0];
    r1._val[2][0] = 0;
    Matrix r3 = Matrix::cross(r1, r2);
    Matrix Hp(4, 4);
    Hp.eye();
    Hp.setMat(r1, 0, 0); Hp.setMat(r2, 0, 1); Hp.setMat(r3, 0, 2);
    dump("r2", r2); dump("r3", r3); dump("Hp", Hp);
    printf("pitch %a l2norm %a mean %a\n", atan2(r3._val[1][0], r3._val[2][0]), Hp.l2norm(), Hp.mean());
    // the rest of the public surface
    Matrix A = H.getMat(1, 0, -1, 2);
    dump("getMat(1,0,-1,2)", A);
    dump("A/A", A / A); dump("A/col", A / A.getMat(0, 1, -1, 1)); dump("A/row", A / A.getMat(2, 0, 2, -1));
    dump("A/2.5", A / 2.5); dump("-A", -A); dump("~A", ~A); dump("A+A-A*3", A + A - A * 3.0);
    Matrix B(3, 5);
    B.setVal(1.5); B.setVal(-2.0, 1, 1, 2, 3); B.setDiag(9.0, 1);
    dump("B", B);
    std::vector<int> idx; idx.push_back(4); idx.push_back(0); idx.push_back(7); idx.push_back(2);
    dump("extractCols", B.extractCols(idx));
    dump("diag(col)", Matrix::diag(v)); dump("diag(row)", Matrix::diag(~v));
    dump("reshape", Matrix::reshape(B, 5, 3));
    FLOAT buf[6];
    B.getData(buf, 1, 2, 2, 4);
    dump("getData", Matrix(2, 3, buf));
    Matrix C(H);
    C.inv();
    dump("H.inv()", C);
    Matrix X = K * Matrix::eye(3);
    X = ~X;
    bool ok = X.solve(K);
    printf("solve %d\n", ok ? 1 : 0);
    dump("solve", X);
    std::cout << P << std::endl << Matrix() << std::endl;
    return 0;
}

/*
 * matrix.h -- small dense double matrix with the public surface the Matcher
 * boundary of the reference relies on (libviso2/src/matrix.h): the class name,
 * the PUBLIC members _val / _m / _n (Matcher::matching reads
 * Tr_delta->_val[i][j], libviso2/src/matcher.cpp:1187-1198; stereomapper reads
 * H._val, stereothread.cpp:203-205) and the usual value semantics and
 * operators.  It is an independent implementation, not the reference's class:
 * what callers at the drop-in boundary and in stereomapper use is provided:
 * construction, copy, element and block access (getData / getMat / setMat / setVal /
 * setDiag / extractCols), eye / diag / reshape / rotMat{X,Y,Z}, + - * / ~ and unary -,
 * l2norm / mean / cross, Gauss-Jordan solve and inverse, operator<<
 * (stereothread.cpp:303-307, maindialog.cpp:396-406, view3d.cpp:93,
 * planeestimation.cpp:98-117).  Not provided: lu / det / svd (no caller on or next to the path).
 * When a translation unit of the reference's applications is built against this
 * tree, THIS header is the one that must be found as "matrix.h" (viso.h and
 * matcher.h here include it); the reference's matrix.cpp is then not linked.
 */
#ifndef MATRIX_H
#define MATRIX_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <stdio.h>

#include <cmath>
#include <iostream>
#include <vector>

typedef double FLOAT;

class Matrix {
public:
    Matrix() : _val(0), _m(0), _n(0) {}
    Matrix(const int32_t m, const int32_t n) : _val(0), _m(0), _n(0) { allocate(m, n); }
    Matrix(const int32_t m, const int32_t n, const FLOAT* val) : _val(0), _m(0), _n(0) {
        allocate(m, n);
        for (int32_t i = 0; i < m; i++)
            for (int32_t j = 0; j < n; j++) _val[i][j] = val[i * n + j];
    }
    Matrix(const Matrix& M) : _val(0), _m(0), _n(0) {
        allocate(M._m, M._n);
        for (int32_t i = 0; i < _m; i++) memcpy(_val[i], M._val[i], _n * sizeof(FLOAT));
    }
    ~Matrix() { release(); }

    Matrix& operator=(const Matrix& M) {
        if (this != &M) {
            if (M._m != _m || M._n != _n) {
                release();
                allocate(M._m, M._n);
            }
            for (int32_t i = 0; i < _m; i++) memcpy(_val[i], M._val[i], _n * sizeof(FLOAT));
        }
        return *this;
    }

    static Matrix eye(const int32_t m) {
        Matrix M(m, m);
        for (int32_t i = 0; i < m; i++) M._val[i][i] = 1;
        return M;
    }
    void zero() {
        for (int32_t i = 0; i < _m; i++) memset(_val[i], 0, _n * sizeof(FLOAT));
    }
    // identity pattern on the existing shape (matrix.h:88)
    void eye() {
        zero();
        for (int32_t i = 0; i < _m && i < _n; i++) _val[i][i] = 1;
    }

    // ---- blocks: rows i1..i2, columns j1..j2 inclusive; -1 = up to the last (matrix.h:67-78)
    void getData(FLOAT* val, int32_t i1 = 0, int32_t j1 = 0, int32_t i2 = -1, int32_t j2 = -1) const {
        last(i2, j2);
        for (int32_t i = i1; i <= i2; i++)
            for (int32_t j = j1; j <= j2; j++) *val++ = _val[i][j];
    }
    Matrix getMat(int32_t i1, int32_t j1, int32_t i2 = -1, int32_t j2 = -1) const {
        last(i2, j2);
        if (i1 < 0 || j1 < 0 || i2 >= _m || j2 >= _n || i2 < i1 || j2 < j1) {
            std::cerr << "ERROR: Cannot get submatrix [" << i1 << ".." << i2 << "] x [" << j1 << ".." << j2 << "]"
                      << " of a (" << _m << "x" << _n << ") matrix." << std::endl;
            exit(0);
        }
        Matrix B(i2 - i1 + 1, j2 - j1 + 1);
        for (int32_t i = 0; i < B._m; i++) memcpy(B._val[i], _val[i1 + i] + j1, B._n * sizeof(FLOAT));
        return B;
    }
    void setMat(const Matrix& B, const int32_t i1, const int32_t j1) {
        if (i1 < 0 || j1 < 0 || i1 + B._m > _m || j1 + B._n > _n) {
            std::cerr << "ERROR: Cannot set submatrix [" << i1 << ".." << i1 + B._m - 1 << "] x [" << j1 << ".."
                      << j1 + B._n - 1 << "]" << " of a (" << _m << "x" << _n << ") matrix." << std::endl;
            exit(0);
        }
        for (int32_t i = 0; i < B._m; i++) memcpy(_val[i1 + i] + j1, B._val[i], B._n * sizeof(FLOAT));
    }
    void setVal(FLOAT s, int32_t i1 = 0, int32_t j1 = 0, int32_t i2 = -1, int32_t j2 = -1) {
        last(i2, j2);
        if (i2 < i1 || j2 < j1) {
            std::cerr << "ERROR in setVal: Indices must be ordered (i1<=i2, j1<=j2)." << std::endl;
            exit(0);
        }
        for (int32_t i = i1; i <= i2; i++)
            for (int32_t j = j1; j <= j2; j++) _val[i][j] = s;
    }
    void setDiag(FLOAT s, int32_t i1 = 0, int32_t i2 = -1) {
        if (i2 == -1) i2 = (_m < _n ? _m : _n) - 1;
        for (int32_t i = i1; i <= i2; i++) _val[i][i] = s;
    }
    // columns idx[0], idx[1], ...; an index past the last column leaves a zero column
    Matrix extractCols(std::vector<int> idx) const {
        Matrix B(_m, (int32_t)idx.size());
        for (int32_t j = 0; j < B._n; j++)
            if (idx[j] < _n)
                for (int32_t i = 0; i < _m; i++) B._val[i][j] = _val[i][idx[j]];
        return B;
    }
    // diagonal matrix from a column or row vector (matrix.h:91)
    static Matrix diag(const Matrix& v) {
        const int32_t n = v._n == 1 ? v._m : v._n;
        if (!((v._m > 1 && v._n == 1) || (v._m == 1 && v._n > 1))) {
            std::cout << "ERROR: Trying to create diagonal matrix from vector of size (" << v._m << "x" << v._n << ")"
                      << std::endl;
            exit(0);
        }
        Matrix D(n, n);
        for (int32_t i = 0; i < n; i++) D._val[i][i] = v._n == 1 ? v._val[i][0] : v._val[0][i];
        return D;
    }
    // same elements in row-major order, new shape (matrix.h:94)
    static Matrix reshape(const Matrix& A, int32_t m, int32_t n) {
        if (A._m * A._n != m * n) {
            std::cerr << "ERROR: Trying to reshape a matrix of size (" << A._m << "x" << A._n << ") to size (" << m
                      << "x" << n << ")" << std::endl;
            exit(0);
        }
        Matrix B(m, n);
        if (m * n > 0) A.getData(B._val[0]);   // both are one row-major block
        return B;
    }
    // rotations about the coordinate axes, right-handed (matrix.h:97-99)
    static Matrix rotMatX(const FLOAT& angle) { return rot(angle, 1, 2); }
    static Matrix rotMatY(const FLOAT& angle) { return rot(angle, 2, 0); }
    static Matrix rotMatZ(const FLOAT& angle) { return rot(angle, 0, 1); }

    Matrix operator+(const Matrix& B) const { return zip(B, +1.0); }
    Matrix operator-(const Matrix& B) const { return zip(B, -1.0); }
    Matrix operator*(const Matrix& B) const {
        Matrix C(_m, B._n);
        if (_n != B._m) {
            std::cerr << "ERROR: Trying to multiply matrices of size (" << _m << "x" << _n << ") and ("
                      << B._m << "x" << B._n << ")" << std::endl;
            exit(0);
        }
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < B._n; j++)
                for (int32_t k = 0; k < _n; k++) C._val[i][j] += _val[i][k] * B._val[k][j];
        return C;
    }
    Matrix operator*(const FLOAT& s) const {
        Matrix C(_m, _n);
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) C._val[i][j] = _val[i][j] * s;
        return C;
    }
    // element-wise quotient; B may also be a column (one divisor per row) or a row (one per
    // column); a zero divisor leaves 0 (matrix.h:106)
    Matrix operator/(const Matrix& B) const {
        const bool same = B._m == _m && B._n == _n, col = B._m == _m && B._n == 1, row = B._n == _n && B._m == 1;
        if (!same && !col && !row) {
            std::cerr << "ERROR: Trying to divide matrices of size (" << _m << "x" << _n << ") and (" << B._m << "x"
                      << B._n << ")" << std::endl;
            exit(0);
        }
        Matrix C(_m, _n);
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) {
                const FLOAT d = same ? B._val[i][j] : (col ? B._val[i][0] : B._val[0][j]);
                if (d != 0) C._val[i][j] = _val[i][j] / d;
            }
        return C;
    }
    Matrix operator/(const FLOAT& s) const {
        if (std::fabs(s) < 1e-20) {
            std::cerr << "ERROR: Trying to divide by zero!" << std::endl;
            exit(0);
        }
        Matrix C(_m, _n);
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) C._val[i][j] = _val[i][j] / s;
        return C;
    }
    Matrix operator-() const {
        Matrix C(_m, _n);
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) C._val[i][j] = -_val[i][j];
        return C;
    }
    // Euclidean / Frobenius norm and mean, summed row by row (matrix.h:110-111)
    FLOAT l2norm() const {
        FLOAT sum = 0;
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) sum += _val[i][j] * _val[i][j];
        return std::sqrt(sum);
    }
    FLOAT mean() const {
        FLOAT sum = 0;
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) sum += _val[i][j];
        return sum / (FLOAT)(_m * _n);
    }
    // a x b of two 3x1 vectors (matrix.h:114)
    static Matrix cross(const Matrix& a, const Matrix& b) {
        if (a._m != 3 || a._n != 1 || b._m != 3 || b._n != 1) {
            std::cerr << "ERROR: Cross product vectors must be of size (3x1)" << std::endl;
            exit(0);
        }
        Matrix c(3, 1);
        for (int32_t k = 0; k < 3; k++) {
            const int32_t p = (k + 1) % 3, q = (k + 2) % 3;
            c._val[k][0] = a._val[p][0] * b._val[q][0] - a._val[q][0] * b._val[p][0];
        }
        return c;
    }
    Matrix operator~() const {
        Matrix C(_n, _m);
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) C._val[j][i] = _val[i][j];
        return C;
    }

    // this = A^-1 * this by Gauss-Jordan elimination with full pivoting
    bool solve(const Matrix& M, FLOAT eps = 1e-20) {
        Matrix A(M);
        if (A._m != A._n || A._m != _m || A._m < 1 || _n < 1) return false;
        const int32_t m = A._m;
        int32_t* used = (int32_t*)calloc(m, sizeof(int32_t));
        bool ok = true;
        for (int32_t it = 0; it < m && ok; it++) {
            FLOAT big = 0;
            int32_t pr = 0, pc = 0;
            for (int32_t j = 0; j < m; j++) {
                if (used[j]) continue;
                for (int32_t k = 0; k < m; k++)
                    if (!used[k] && std::fabs(A._val[j][k]) >= big) {
                        big = std::fabs(A._val[j][k]);
                        pr = j;
                        pc = k;
                    }
            }
            used[pc] = 1;
            if (pr != pc) {
                for (int32_t l = 0; l < m; l++) std::swap(A._val[pr][l], A._val[pc][l]);
                for (int32_t l = 0; l < _n; l++) std::swap(_val[pr][l], _val[pc][l]);
            }
            if (std::fabs(A._val[pc][pc]) < eps) {
                ok = false;
                break;
            }
            const FLOAT inv = 1.0 / A._val[pc][pc];
            A._val[pc][pc] = 1.0;
            for (int32_t l = 0; l < m; l++) A._val[pc][l] *= inv;
            for (int32_t l = 0; l < _n; l++) _val[pc][l] *= inv;
            for (int32_t r = 0; r < m; r++) {
                if (r == pc) continue;
                const FLOAT f = A._val[r][pc];
                A._val[r][pc] = 0.0;
                for (int32_t l = 0; l < m; l++) A._val[r][l] -= A._val[pc][l] * f;
                for (int32_t l = 0; l < _n; l++) _val[r][l] -= _val[pc][l] * f;
            }
        }
        free(used);
        return ok;
    }
    static Matrix inv(const Matrix& M) {
        Matrix B = eye(M._m);
        B.solve(M);
        return B;
    }
    bool inv() {
        Matrix B = eye(_m);
        if (!B.solve(*this)) return false;
        *this = B;
        return true;
    }

    // "%12.7f " per element, rows separated by a line break (matrix.cpp:1157-1181)
    friend std::ostream& operator<<(std::ostream& out, const Matrix& M) {
        if (M._m == 0 || M._n == 0) return out << "[empty matrix]";
        char buf[64];
        for (int32_t i = 0; i < M._m; i++) {
            for (int32_t j = 0; j < M._n; j++) {
                snprintf(buf, sizeof(buf), "%12.7f ", M._val[i][j]);
                out << buf;
            }
            if (i < M._m - 1) out << std::endl;
        }
        return out;
    }

    // direct data access (public in the reference as well)
    FLOAT** _val;
    int32_t _m;  // rows
    int32_t _n;  // columns

private:
    void last(int32_t& i2, int32_t& j2) const {
        if (i2 == -1) i2 = _m - 1;
        if (j2 == -1) j2 = _n - 1;
    }
    // rotation by `angle` in the plane of axes (a, b): R[a][a] = R[b][b] = cos, R[b][a] = sin
    static Matrix rot(const FLOAT& angle, int32_t a, int32_t b) {
        const FLOAT sn = std::sin(angle), cs = std::cos(angle);
        Matrix R(3, 3);
        R._val[3 - a - b][3 - a - b] = 1;
        R._val[a][a] = cs;
        R._val[a][b] = -sn;
        R._val[b][a] = sn;
        R._val[b][b] = cs;
        return R;
    }
    void allocate(const int32_t m, const int32_t n) {
        _m = m;
        _n = n;
        if (m <= 0 || n <= 0) {
            _val = 0;
            return;
        }
        _val = (FLOAT**)malloc(m * sizeof(FLOAT*));
        _val[0] = (FLOAT*)calloc((size_t)m * n, sizeof(FLOAT));
        for (int32_t i = 1; i < m; i++) _val[i] = _val[i - 1] + n;
    }
    void release() {
        if (_val) {
            free(_val[0]);
            free(_val);
        }
        _val = 0;
        _m = _n = 0;
    }
    Matrix zip(const Matrix& B, FLOAT sign) const {
        Matrix C(_m, _n);
        if (B._m != _m || B._n != _n) {
            std::cerr << "ERROR: Trying to add matrices of size (" << _m << "x" << _n << ") and (" << B._m
                      << "x" << B._n << ")" << std::endl;
            exit(0);
        }
        for (int32_t i = 0; i < _m; i++)
            for (int32_t j = 0; j < _n; j++) C._val[i][j] = _val[i][j] + sign * B._val[i][j];
        return C;
    }
};

#endif

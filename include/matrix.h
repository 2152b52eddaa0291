/*
 * matrix.h -- small dense double matrix with the public surface the Matcher
 * boundary of the reference relies on (libviso2/src/matrix.h): the class name,
 * the PUBLIC members _val / _m / _n (Matcher::matching reads
 * Tr_delta->_val[i][j], libviso2/src/matcher.cpp:1187-1198; stereomapper reads
 * H._val, stereothread.cpp:203-205) and the usual value semantics and
 * operators.  It is an independent implementation, not the reference's class:
 * only what callers at the drop-in boundary need is provided (construction,
 * copy, element access, eye/zero, + - * ~, Gauss-Jordan solve and inverse).
 */
#ifndef MATRIX_H
#define MATRIX_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <iostream>

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

    // direct data access (public in the reference as well)
    FLOAT** _val;
    int32_t _m;  // rows
    int32_t _n;  // columns

private:
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

// oracle/ref_shim/mini_dense.hpp -- TEST INFRASTRUCTURE ONLY.
// A from-scratch dense-matrix stand-in with the Eigen SPELLINGS that the local-BA row loops of
// /root/reference/src/mapHandler.cpp use (:1358-1431, :1436-1540, :1587-1666, :1668-1772; and the pose-only Gauss-Newton
// loops :3330-3367, :3370-3412 and their twins in computeRelativePoseRobustGN): fixed and dynamic double
// matrices, (i) / (i,j), block / head / tail as assignable views, transpose, norm, Zero, the comma initialiser,
// + - * / with matrices and scalars.  It lets those loops be compiled TEXTUALLY from where they lie (oracle/
// ref_extract_lba.py cuts them into oracle/_ref/*.inc at build time) so that the oracle's restatement of the row
// algebra and of the H / g accumulation can be checked against the reference's own source text.  NOT Eigen: every
// operation is a plain loop (products sum k = 0, 1, 2, ... in order), so rounding may differ from a real Eigen build
// in the last bits -- the check built on it is a 1e-12-relative one, about algebra, signs and indices.
#ifndef PLSLAM_ORACLE_REF_SHIM_MINI_DENSE
#define PLSLAM_ORACLE_REF_SHIM_MINI_DENSE
#include <cmath>
#include <stdexcept>
#include <vector>

namespace mini {
struct Dyn;
struct Block {                       // assignable view
    Dyn* m;
    int i0, j0, h, w;
    Block& operator=(const Dyn& o);
    Block& operator+=(const Dyn& o);
    Block& operator=(const Block& o);
    Block head(int n) const { Block b = {m, i0, j0, w == 1 ? n : 1, w == 1 ? 1 : n}; return b; }
};
struct Comma {
    Dyn* m;
    int k;
    Comma operator,(double x);
    Comma operator,(const Dyn& x);
};
struct Dyn {
    int r, c;
    std::vector<double> v;           // row-major
    Dyn() : r(0), c(0) {}
    Dyn(int r_, int c_) : r(r_), c(c_), v((size_t)r_ * c_, 0.0) {}
    Dyn(const Block& b) : r(b.h), c(b.w), v((size_t)b.h * b.w) {
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) v[(size_t)i * c + j] = b.m->v[(size_t)(b.i0 + i) * b.m->c + b.j0 + j];
    }
    double& operator()(int i) { return v.at(i); }
    const double& operator()(int i) const { return v.at(i); }
    double& operator()(int i, int j) { return v.at((size_t)i * c + j); }
    const double& operator()(int i, int j) const { return v.at((size_t)i * c + j); }
    Block block(int i, int j, int h, int w) {
        if (i < 0 || j < 0 || i + h > r || j + w > c) throw std::out_of_range("mini::block");
        Block b = {this, i, j, h, w};
        return b;
    }
    Block col(int j) { return block(0, j, r, 1); }
    const Dyn& sparseView() const { return *this; }      // (H.sparseView(), src/mapHandler.cpp:1555: the dense matrix itself)
    Dyn& operator+=(const Dyn& o) {
        if (o.r != r || o.c != c) throw std::invalid_argument("mini::+= : shape");
        for (size_t k = 0; k < v.size(); ++k) v[k] += o.v[k];
        return *this;
    }
    Block head(int n) { return c == 1 ? block(0, 0, n, 1) : block(0, 0, 1, n); }
    Block tail(int n) { return c == 1 ? block(r - n, 0, n, 1) : block(0, c - n, 1, n); }
    Dyn transpose() const {
        Dyn t(c, r);
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < c; ++j) t.v[(size_t)j * r + i] = v[(size_t)i * c + j];
        return t;
    }
    double norm() const {
        double s = 0.0;
        for (size_t k = 0; k < v.size(); ++k) s += v[k] * v[k];
        return std::sqrt(s);
    }
    Comma operator<<(double x) { v.at(0) = x; Comma cm = {this, 1}; return cm; }
    Comma operator<<(const Dyn& x) { Comma cm = {this, 0}; return (cm, x); }
    Dyn normalized() const { Dyn o = *this; const double n = norm(); for (size_t k = 0; k < o.v.size(); ++k) o.v[k] = v[k] / n; return o; }
};
inline Comma Comma::operator,(const Dyn& x) {
    Comma cm = {m, k};
    for (size_t i = 0; i < x.v.size(); ++i) { m->v.at(cm.k) = x.v[i]; ++cm.k; }
    return cm;
}
inline Comma Comma::operator,(double x) { m->v.at(k) = x; Comma cm = {m, k + 1}; return cm; }
inline Block& Block::operator=(const Dyn& o) {
    const bool same = o.r == h && o.c == w, flip = o.r == w && o.c == h && (h == 1 || w == 1);
    if (!same && !flip) throw std::invalid_argument("mini::Block = : shape");
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) m->v[(size_t)(i0 + i) * m->c + j0 + j] = same ? o.v[(size_t)i * o.c + j] : o.v[(size_t)j * o.c + i];
    return *this;
}
inline Block& Block::operator=(const Block& o) { return *this = Dyn(o); }
inline Block& Block::operator+=(const Dyn& o) {
    const bool same = o.r == h && o.c == w, flip = o.r == w && o.c == h && (h == 1 || w == 1);
    if (!same && !flip) throw std::invalid_argument("mini::Block += : shape");
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) m->v[(size_t)(i0 + i) * m->c + j0 + j] += same ? o.v[(size_t)i * o.c + j] : o.v[(size_t)j * o.c + i];
    return *this;
}
inline Dyn operator*(const Dyn& a, const Dyn& b) {
    if (a.c != b.r) throw std::invalid_argument("mini::* : shape");
    Dyn o(a.r, b.c);
    for (int i = 0; i < a.r; ++i)
        for (int j = 0; j < b.c; ++j) {
            double s = a.v[(size_t)i * a.c] * b.v[j];
            for (int k = 1; k < a.c; ++k) s += a.v[(size_t)i * a.c + k] * b.v[(size_t)k * b.c + j];
            o.v[(size_t)i * o.c + j] = s;
        }
    return o;
}
inline Dyn operator*(const Dyn& a, double s) { Dyn o = a; for (size_t k = 0; k < o.v.size(); ++k) o.v[k] = a.v[k] * s; return o; }
inline Dyn operator*(double s, const Dyn& a) { return a * s; }
inline Dyn operator/(const Dyn& a, double s) { Dyn o = a; for (size_t k = 0; k < o.v.size(); ++k) o.v[k] = a.v[k] / s; return o; }
inline Dyn addsub(const Dyn& a, const Dyn& b, double sg) {
    if (a.r != b.r || a.c != b.c) throw std::invalid_argument("mini::+- : shape");
    Dyn o = a;
    for (size_t k = 0; k < o.v.size(); ++k) o.v[k] = a.v[k] + sg * b.v[k];
    return o;
}
inline Dyn operator+(const Dyn& a, const Dyn& b) { return addsub(a, b, 1.0); }
inline Dyn operator-(const Dyn& a, const Dyn& b) { return addsub(a, b, -1.0); }

template <int R, int C>
struct Fixed : Dyn {
    Fixed() : Dyn(R, C) {}
    Fixed(const Dyn& d) : Dyn(R, C) { assign(d); }
    Fixed(const Block& b) : Dyn(R, C) { assign(Dyn(b)); }
    Fixed& operator=(const Dyn& d) { assign(d); return *this; }
    static Fixed Zero() { return Fixed(); }
private:
    void assign(const Dyn& d) {      // Eigen lets a row vector be assigned to a column vector and vice versa
        if (d.r == R && d.c == C) v = d.v;
        else if (d.r == C && d.c == R && (R == 1 || C == 1)) v = d.v;
        else throw std::invalid_argument("mini::Fixed = : shape");
    }
};
struct MatrixX : Dyn {
    MatrixX() {}
    MatrixX(int r_, int c_) : Dyn(r_, c_) {}
    MatrixX(const Dyn& d) : Dyn(d) {}
    static MatrixX Zero(int r_, int c_) { return MatrixX(r_, c_); }
    static MatrixX Zero(int n) { return MatrixX(n, 1); }
};
struct SparseLike : Dyn {            // SparseMatrix / SparseVector spelt through coeffRef only; dense underneath
    SparseLike() {}
    SparseLike(int r_, int c_) : Dyn(r_, c_) {}
    explicit SparseLike(int n) : Dyn(n, 1) {}
    double& coeffRef(int i, int j) { return (*this)(i, j); }
    double& coeffRef(int i) { return (*this)(i); }
};
struct Vector6i {
    int v[6];
    int operator()(int i) const { return v[i]; }
    int& operator()(int i) { return v[i]; }
};
}  // namespace mini
#endif

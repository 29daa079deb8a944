// mini_dense.h -- a minimal stand-in for the few Eigen facilities that the
// reference's hot-path headers touch, so that those UNMODIFIED headers can be
// compiled where Eigen is not installed (oracle/_ref, test infrastructure).
//
// This is our own code, written against Eigen's documented behaviour; it is not
// derived from Eigen's sources.  Semantics that matter for parity:
//   * v / s is a true per-coefficient division (never * (1/s));
//   * dot() and squaredNorm() sum products in index order;
//   * normalized() divides by sqrt(squaredNorm()) when that is > 0;
//   * lpNorm<Infinity>() is max |x_i|;
//   * operator== is exact coefficient-wise equality.
// Dense LU inverse(), PolynomialSolver etc. are only here to let unreachable
// inline code parse; PolynomialSolver aborts if it is ever executed.
#ifndef MPL_ORACLE_MINI_DENSE_H
#define MPL_ORACLE_MINI_DENSE_H

#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>
#include <ostream>
#include <vector>

namespace Eigen {

const int Dynamic = -1;
const int Infinity = -1;
enum { Affine = 2 };
enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };
enum ComputationInfo { Success = 0, NumericalIssue = 1 };

template <typename T>
using aligned_allocator = std::allocator<T>;

template <typename Derived>
class MatrixBase {};

template <typename T, int R, int C>
class Matrix;

template <typename M>
class CommaInit {
 public:
  CommaInit(M &m, typename M::Scalar first) : m_(m), k_(0) { put(first); }
  CommaInit &operator,(typename M::Scalar v) { put(v); return *this; }

 private:
  void put(typename M::Scalar v) {
    const int cols = m_.cols();
    m_(k_ / cols, k_ % cols) = v;  // row-major fill order, like Eigen
    k_++;
  }
  M &m_;
  int k_;
};

template <typename T, int R, int C, bool Dyn = (R == Dynamic || C == Dynamic)>
struct Storage {
  std::array<T, (size_t)(R * C)> d;
  Storage() { d.fill(T(0)); }
  void resize(int, int) {}
  int rows() const { return R; }
  int cols() const { return C; }
};
template <typename T, int R, int C>
struct Storage<T, R, C, true> {
  std::vector<T> d;
  int r = (R == Dynamic ? 0 : R), c = (C == Dynamic ? 0 : C);
  void resize(int rr, int cc) { r = rr; c = cc; d.assign((size_t)rr * cc, T(0)); }
  int rows() const { return r; }
  int cols() const { return c; }
};

template <typename T, int R, int C = 1>
class Matrix : public MatrixBase<Matrix<T, R, C>> {
 public:
  typedef T Scalar;
  typedef Matrix PlainObject;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C };
  static constexpr bool kDyn = (R == Dynamic || C == Dynamic);

  Matrix() {}
  // size constructor for dynamic vectors, value constructors for fixed ones
  explicit Matrix(int n) {
    if (kDyn) s_.resize(R == Dynamic ? n : R, C == Dynamic ? (R == Dynamic ? 1 : n) : C);
  }
  Matrix(T a, T b) {
    if (kDyn) s_.resize((int)a, (int)b);
    else { (*this)(0) = a; (*this)(1) = b; }
  }
  Matrix(T a, T b, T c) { (*this)(0) = a; (*this)(1) = b; (*this)(2) = c; }
  Matrix(T a, T b, T c, T d) { (*this)(0) = a; (*this)(1) = b; (*this)(2) = c; (*this)(3) = d; }
  // fixed <- dynamic / different static type with equal size
  template <int R2, int C2>
  Matrix(const Matrix<T, R2, C2> &o) {
    if (kDyn) s_.resize(o.rows(), o.cols());
    for (int i = 0; i < o.size(); i++) s_.d[(size_t)i] = o.data()[i];
  }

  int rows() const { return s_.rows(); }
  int cols() const { return s_.cols(); }
  int size() const { return rows() * cols(); }
  const T *data() const { return s_.d.data(); }
  T *data() { return s_.d.data(); }

  // column-major storage
  T &operator()(int i) { return s_.d[(size_t)i]; }
  const T &operator()(int i) const { return s_.d[(size_t)i]; }
  T &operator[](int i) { return s_.d[(size_t)i]; }
  const T &operator[](int i) const { return s_.d[(size_t)i]; }
  T &operator()(int i, int j) { return s_.d[(size_t)(i + j * rows())]; }
  const T &operator()(int i, int j) const { return s_.d[(size_t)(i + j * rows())]; }

  CommaInit<Matrix> operator<<(T first) { return CommaInit<Matrix>(*this, first); }

  static Matrix Zero() { return Matrix(); }
  static Matrix Zero(int r, int c) { Matrix m; m.s_.resize(r, c); return m; }
  static Matrix Constant(T v) { Matrix m; for (int i = 0; i < m.size(); i++) m(i) = v; return m; }
  static Matrix Identity(int r, int c) {
    Matrix m = Zero(r, c);
    for (int i = 0; i < r && i < c; i++) m(i, i) = T(1);
    return m;
  }

  Matrix operator+(const Matrix &o) const { Matrix m(*this); for (int i = 0; i < size(); i++) m(i) = (*this)(i) + o(i); return m; }
  Matrix operator-(const Matrix &o) const { Matrix m(*this); for (int i = 0; i < size(); i++) m(i) = (*this)(i) - o(i); return m; }
  Matrix operator-() const { Matrix m(*this); for (int i = 0; i < size(); i++) m(i) = -(*this)(i); return m; }
  Matrix operator*(T s) const { Matrix m(*this); for (int i = 0; i < size(); i++) m(i) = (*this)(i) * s; return m; }
  Matrix operator/(T s) const { Matrix m(*this); for (int i = 0; i < size(); i++) m(i) = (*this)(i) / s; return m; }
  friend Matrix operator*(T s, const Matrix &a) { Matrix m(a); for (int i = 0; i < a.size(); i++) m(i) = s * a(i); return m; }
  Matrix &operator+=(const Matrix &o) { for (int i = 0; i < size(); i++) (*this)(i) += o(i); return *this; }
  Matrix &operator-=(const Matrix &o) { for (int i = 0; i < size(); i++) (*this)(i) -= o(i); return *this; }
  Matrix &operator*=(T s) { for (int i = 0; i < size(); i++) (*this)(i) *= s; return *this; }
  Matrix &operator/=(T s) { for (int i = 0; i < size(); i++) (*this)(i) /= s; return *this; }

  template <int C2>
  Matrix<T, R, C2> operator*(const Matrix<T, C, C2> &o) const {
    Matrix<T, R, C2> m;
    for (int i = 0; i < rows(); i++)
      for (int j = 0; j < o.cols(); j++) {
        T acc = T(0);
        for (int k = 0; k < cols(); k++) acc += (*this)(i, k) * o(k, j);
        m(i, j) = acc;
      }
    return m;
  }

  bool operator==(const Matrix &o) const {
    for (int i = 0; i < size(); i++)
      if (!((*this)(i) == o(i))) return false;
    return true;
  }
  bool operator!=(const Matrix &o) const { return !(*this == o); }

  // Summation order of a 3-vector's dot / squaredNorm: index order, (x0 y0 + x1 y1) + x2 y2.  That is the tree of Eigen
  // 3.3's redux on an SSE2 build with EIGEN_UNALIGNED_VECTORIZE (its default): one Packet2d of products reduced, then the
  // scalar tail added (Redux.h, LinearVectorizedTraversal + CompleteUnrolling).  A build without vectorisation (Eigen 3.2,
  // or -DEIGEN_DONT_VECTORIZE) unrolls the scalar redux in halves instead: x0 y0 + (x1 y1 + x2 y2).  The reference pins
  // no Eigen version, so either is "the reference"; the two differ by at most two units in the last
  // place of the norm (<= 4.5e-16 relative), and on the hot path only env_map.h:116's vel.norm() on 3D potential
  // maps with gradient_weight != 0 sees a 3-vector -- a COST, for which north_star allows 1e-6.  2-vectors (validate_yaw,
  // the heading cost) have one tree only.  tests/test_oracle_known_answers.py bounds the difference.
  T dot(const Matrix &o) const { T acc = (*this)(0) * o(0); for (int i = 1; i < size(); i++) acc += (*this)(i) * o(i); return acc; }
  T squaredNorm() const { return dot(*this); }
  T norm() const { return std::sqrt(squaredNorm()); }
  Matrix normalized() const {
    const T z = squaredNorm();
    if (z > T(0)) return *this / std::sqrt(z);
    return *this;
  }
  template <int P>
  T lpNorm() const {
    static_assert(P == Infinity, "only lpNorm<Infinity> is provided");
    T m = T(0);
    for (int i = 0; i < size(); i++) { T a = std::abs((*this)(i)); if (a > m) m = a; }
    return m;
  }
  template <int N>
  Matrix<T, N, 1> topRows() const { Matrix<T, N, 1> m; for (int i = 0; i < N; i++) m(i) = (*this)(i); return m; }
  template <typename U>
  Matrix<U, R, C> cast() const { Matrix<U, R, C> m; for (int i = 0; i < size(); i++) m(i) = (U)(*this)(i); return m; }
  Matrix<T, C, R> transpose() const {
    Matrix<T, C, R> m;
    for (int i = 0; i < rows(); i++) for (int j = 0; j < cols(); j++) m(j, i) = (*this)(i, j);
    return m;
  }
  Matrix inverse() const {  // Gauss-Jordan with partial pivoting; square only
    const int n = rows();
    Matrix a(*this), inv = Matrix::Zero(n, n);
    for (int i = 0; i < n; i++) inv(i, i) = T(1);
    for (int col = 0; col < n; col++) {
      int piv = col;
      for (int r = col + 1; r < n; r++) if (std::abs(a(r, col)) > std::abs(a(piv, col))) piv = r;
      for (int j = 0; j < n; j++) { std::swap(a(col, j), a(piv, j)); std::swap(inv(col, j), inv(piv, j)); }
      const T d = a(col, col);
      for (int j = 0; j < n; j++) { a(col, j) /= d; inv(col, j) /= d; }
      for (int r = 0; r < n; r++) {
        if (r == col) continue;
        const T f = a(r, col);
        for (int j = 0; j < n; j++) { a(r, j) -= f * a(col, j); inv(r, j) -= f * inv(col, j); }
      }
    }
    return inv;
  }

 private:
  Storage<T, R, C> s_;
};

template <typename T, int R, int C>
std::ostream &operator<<(std::ostream &os, const Matrix<T, R, C> &m) {
  for (int i = 0; i < m.rows(); i++) {
    for (int j = 0; j < m.cols(); j++) os << (j ? " " : "") << m(i, j);
    if (i + 1 < m.rows()) os << "\n";
  }
  return os;
}

typedef Matrix<double, Dynamic, 1> VectorXd;

template <typename T, int Dim, int Mode>
class Transform {};

// Declarations only: used inside function templates of math.h that the hot
// path never instantiates.
template <typename M> class JacobiSVD;
template <typename M> class LLT;
template <typename M> class LDLT;

template <typename T, int Deg>
class PolynomialSolver {
 public:
  typedef Matrix<std::complex<T>, Dynamic, 1> RootsType;
  template <typename V>
  void compute(const V &) {
    fprintf(stderr, "mini_dense: PolynomialSolver is outside the hot path and not provided\n");
    abort();
  }
  const RootsType &roots() const { return r_; }

 private:
  RootsType r_;
};

}  // namespace Eigen
#endif

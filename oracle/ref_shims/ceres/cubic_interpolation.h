// Stand-in for ceres/cubic_interpolation.h: CubicHermiteSpline (Catmull-Rom through p1, p2) and
// BiCubicInterpolator (four row splines, then one column spline of the values and one of the row
// derivatives), written from the published header.  Ours; TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REF_SHIMS_CERES_CUBIC_INTERPOLATION_H_
#define ORACLE_REF_SHIMS_CERES_CUBIC_INTERPOLATION_H_
#include <cmath>
namespace ceres {

template <int kDataDimension>
void CubicHermiteSpline(const double* p0, const double* p1, const double* p2, const double* p3,
                        const double x, double* f, double* dfdx) {
  for (int i = 0; i < kDataDimension; ++i) {
    const double a = 0.5 * (-p0[i] + 3.0 * p1[i] - 3.0 * p2[i] + p3[i]);
    const double b = 0.5 * (2.0 * p0[i] - 5.0 * p1[i] + 4.0 * p2[i] - p3[i]);
    const double c = 0.5 * (-p0[i] + p2[i]);
    const double d = p1[i];
    // Use Horner's rule to evaluate the function value and its derivative.
    if (f != nullptr) f[i] = d + x * (c + x * (b + x * a));
    if (dfdx != nullptr) dfdx[i] = c + x * (2.0 * b + 3.0 * a * x);
  }
}

template <typename Grid>
class BiCubicInterpolator {
 public:
  explicit BiCubicInterpolator(const Grid& grid) : grid_(grid) {}

  void Evaluate(double r, double c, double* f, double* dfdr, double* dfdc) const {
    enum { D = Grid::DATA_DIMENSION };
    const int row = static_cast<int>(std::floor(r));
    const int col = static_cast<int>(std::floor(c));
    double p0[D], p1[D], p2[D], p3[D];
    double fr[4][D], dfr[4][D];
    for (int k = 0; k < 4; ++k) {
      grid_.GetValue(row - 1 + k, col - 1, p0);
      grid_.GetValue(row - 1 + k, col, p1);
      grid_.GetValue(row - 1 + k, col + 1, p2);
      grid_.GetValue(row - 1 + k, col + 2, p3);
      CubicHermiteSpline<D>(p0, p1, p2, p3, c - col, fr[k], dfr[k]);
    }
    CubicHermiteSpline<D>(fr[0], fr[1], fr[2], fr[3], r - row, f, dfdr);
    if (dfdc != nullptr)
      CubicHermiteSpline<D>(dfr[0], dfr[1], dfr[2], dfr[3], r - row, dfdc, nullptr);
  }
  void Evaluate(const double& r, const double& c, double* f) const {
    Evaluate(r, c, f, nullptr, nullptr);
  }
  template <typename JetT>
  void Evaluate(const JetT& r, const JetT& c, JetT* f) const {
    enum { D = Grid::DATA_DIMENSION };
    double frc[D], dfdr[D], dfdc[D];
    Evaluate(r.a, c.a, frc, dfdr, dfdc);
    for (int i = 0; i < D; ++i) {
      f[i].a = frc[i];
      for (int k = 0; k < JetT::DIMENSION; ++k) f[i].v[k] = dfdr[i] * r.v[k] + dfdc[i] * c.v[k];
    }
  }

 private:
  const Grid& grid_;
};

}  // namespace ceres
#endif  // ORACLE_REF_SHIMS_CERES_CUBIC_INTERPOLATION_H_

// libtfc: error reporting, version, and the host-side real Clebsch-Gordan tables.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/tfc.h"

static thread_local char g_err[512] = "";

int tfc_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* tfc_last_error(void) { return g_err; }

extern "C" int tfc_version(void) { return 100; }

// ---------------------------------------------------------------------------------------------
// Real Clebsch-Gordan tensor, e3nn >= 0.4 convention (o3.wigner_3j as used by o3.TensorProduct,
// reference call site eigenn/nn/utils.py:231-238): complex Condon-Shortley coefficients rotated
// to the real basis (y polar axis) and scaled to unit Frobenius norm.
// ---------------------------------------------------------------------------------------------
static double fact(int n) {
    double f = 1.0;
    for (int i = 2; i <= n; ++i) f *= i;
    return f;
}

static double su2_cg(int j1, int m1, int j2, int m2, int j3, int m3) {
    if (m3 != m1 + m2) return 0.0;
    const int vmin = std::max(std::max(-j1 + j2 + m3, -j1 + m1), 0);
    const int vmax = std::min(std::min(j2 + j3 + m1, j3 - j1 + j2), j3 + m3);
    const double c = std::sqrt((2.0 * j3 + 1.0) * fact(j3 + j1 - j2) * fact(j3 - j1 + j2) * fact(j1 + j2 - j3) *
                               fact(j3 + m3) * fact(j3 - m3) /
                               (fact(j1 + j2 + j3 + 1) * fact(j1 - m1) * fact(j1 + m1) * fact(j2 - m2) *
                                fact(j2 + m2)));
    double s = 0.0;
    for (int v = vmin; v <= vmax; ++v) {
        const double sign = ((v + j2 + m2) & 1) ? -1.0 : 1.0;
        s += sign * fact(j2 + j3 + m1 - v) * fact(j1 - m1 + v) /
             (fact(v) * fact(j3 - j1 + j2 - v) * fact(j3 + m3 - v) * fact(v + j1 - j2 - m3));
    }
    return c * s;
}

typedef std::complex<double> cplx;

// Q[row = complex m][col = real index]
static std::vector<cplx> real_to_complex(int l) {
    const int d = 2 * l + 1;
    std::vector<cplx> q((size_t)d * d, cplx(0, 0));
    const double s = 1.0 / std::sqrt(2.0);
    for (int m = -l; m < 0; ++m) {
        q[(size_t)(l + m) * d + (l - m)] = cplx(s, 0);
        q[(size_t)(l + m) * d + (l + m)] = cplx(0, -s);
    }
    q[(size_t)l * d + l] = cplx(1, 0);
    for (int m = 1; m <= l; ++m) {
        const double sg = (m & 1) ? -1.0 : 1.0;
        q[(size_t)(l + m) * d + (l + m)] = cplx(sg * s, 0);
        q[(size_t)(l + m) * d + (l - m)] = cplx(0, sg * s);
    }
    cplx f(1, 0);
    for (int i = 0; i < l; ++i) f *= cplx(0, -1);  // (-i)^l
    for (auto& v : q) v *= f;
    return q;
}

extern "C" int tfc_real_cg(int l1, int l2, int l3, double* out) {
    if (!out) return tfc_set_error(TFC_EINVAL, "tfc_real_cg: null output");
    if (l1 < 0 || l2 < 0 || l3 < 0 || l1 > 6 || l2 > 6 || l3 > 6)
        return tfc_set_error(TFC_EUNSUPPORTED, "tfc_real_cg: l outside 0..6");
    if (std::abs(l1 - l2) > l3 || l3 > l1 + l2)
        return tfc_set_error(TFC_EINVAL, "tfc_real_cg: (%d,%d,%d) violates the triangle rule", l1, l2, l3);
    const int d1 = 2 * l1 + 1, d2 = 2 * l2 + 1, d3 = 2 * l3 + 1;
    std::vector<double> c((size_t)d1 * d2 * d3, 0.0);
    for (int a = -l1; a <= l1; ++a)
        for (int b = -l2; b <= l2; ++b)
            if (std::abs(a + b) <= l3)
                c[((size_t)(l1 + a) * d2 + (l2 + b)) * d3 + (l3 + a + b)] = su2_cg(l1, a, l2, b, l3, a + b);
    const std::vector<cplx> q1 = real_to_complex(l1), q2 = real_to_complex(l2), q3 = real_to_complex(l3);
    // out[j][l][m] = sum_{i,k,n} q1[i][j] q2[k][l] conj(q3[n][m]) c[i][k][n]
    double norm2 = 0.0, imag_max = 0.0;
    for (int j = 0; j < d1; ++j)
        for (int l = 0; l < d2; ++l)
            for (int m = 0; m < d3; ++m) {
                cplx acc(0, 0);
                for (int i = 0; i < d1; ++i) {
                    const cplx a = q1[(size_t)i * d1 + j];
                    if (a == cplx(0, 0)) continue;
                    for (int k = 0; k < d2; ++k) {
                        const cplx b = q2[(size_t)k * d2 + l];
                        if (b == cplx(0, 0)) continue;
                        for (int n = 0; n < d3; ++n) {
                            const double v = c[((size_t)i * d2 + k) * d3 + n];
                            if (v == 0.0) continue;
                            acc += a * b * std::conj(q3[(size_t)n * d3 + m]) * v;
                        }
                    }
                }
                imag_max = std::max(imag_max, std::fabs(acc.imag()));
                double r = acc.real();
                if (std::fabs(r) < 1e-14) r = 0.0;
                out[((size_t)j * d2 + l) * d3 + m] = r;
                norm2 += r * r;
            }
    if (imag_max > 1e-9) return tfc_set_error(TFC_EINVAL, "tfc_real_cg: imaginary part %g", imag_max);
    const double inv = 1.0 / std::sqrt(norm2);
    for (size_t i = 0; i < (size_t)d1 * d2 * d3; ++i) out[i] *= inv;
    return TFC_OK;
}

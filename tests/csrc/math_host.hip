// Test-only harness: exposes the host instantiation of the __host__ __device__ numerics in
// filter_functions_amd/csrc/ffk_math.h through a C ABI so that tests/test_math_host.py can
// compare them with NumPy on a machine without a GPU.  Not part of the product library.
#include "ffk_math.h"

extern "C" {

void ffk_host_sincos(long n, const double* x, double* s, double* c) {
    for (long i = 0; i < n; ++i) ffk::sincos_pi(x[i], &s[i], &c[i]);
}

void ffk_host_first_order_integral(long n, const double* omega, const double* dE, double dt,
                                   double* out) {
    for (long i = 0; i < n; ++i) {
        ffk::cplx v = ffk::first_order_integral(omega[i], dE[i], dt);
        out[2*i] = v.re;
        out[2*i + 1] = v.im;
    }
}

void ffk_host_first_order_integral_aa(long n, const double* omega, const double* dE, double dt,
                                      double* out) {
    for (long i = 0; i < n; ++i) {
        double sa, ca, sb, cb;
        ffk::sincos_pi(0.5*(omega[i]*dt), &sa, &ca);
        ffk::sincos_pi(0.5*(dE[i]*dt), &sb, &cb);
        ffk::cplx v = ffk::first_order_integral_aa(omega[i], dE[i], dt, sa, ca, sb, cb);
        out[2*i] = v.re;
        out[2*i + 1] = v.im;
    }
}

}  // extern "C"

// ffk_math.h -- scalar FP64 building blocks shared by every kernel of the filter-function path.
//
// All functions are __host__ __device__ so that tests/ can compile them for the host (a
// test-only harness, tests/csrc/) and check them against NumPy without a GPU; the product
// library only ever runs them on the device.
//
// Rounding contract (DESIGN.md "Numerics"): the reference evaluates sin/cos at the *rounded*
// arguments fl(omega*t_g) (numeric.py:865) and fl(fl(omega + dE)*dt) (numeric.py:155-163).
// For omega*t up to ~1e6 the rounding of the argument itself is worth ~1e-10 relative in the
// result, so the kernels form exactly the same double-precision arguments and only the
// sin/cos implementation (<= ~1 ulp) differs from NumPy's.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

#define FFK_HD __host__ __device__ __forceinline__

namespace ffk {

struct cplx {
    double re, im;
};

FFK_HD cplx cmul(cplx a, cplx b) {
    return {fma(a.re, b.re, -(a.im*b.im)), fma(a.re, b.im, a.im*b.re)};
}
// acc += a*b
FFK_HD void cmac(cplx& acc, cplx a, cplx b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(a.im, b.re, acc.im);
}
// acc += conj(a)*b
FFK_HD void cmac_conj(cplx& acc, cplx a, cplx b) {
    acc.re = fma(a.re, b.re, acc.re);
    acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im);
    acc.im = fma(-a.im, b.re, acc.im);
}

// ---------------------------------------------------------------------------------------
// sincos for |x| < ~2^50: 3-term Cody-Waite reduction with FMA (pi/2 = P1 + P2 + P3 carries
// ~160 bits, so the reduction error stays ~1e-16 absolute for every k that is an exact
// double), then the classic minimax kernels on [-pi/4, pi/4] (13th/14th degree; the
// coefficients are the widely published fdlibm k_sin/k_cos constants).  Max error observed
// against NumPy on random arguments up to 1e8: 1 ulp (tests/test_math_host.py).
// round-to-nearest and the quadrant come from the 1.5*2^52 "magic number" addition: the low
// mantissa bits of t hold the integer k.  NaN/Inf give NaN.
// ---------------------------------------------------------------------------------------
// FFK_PIN(c): the constant is materialised (two v_mov_b32) where it is used.  Inside a loop the
// compiler otherwise hoists every polynomial coefficient into a register that stays live across
// the whole loop body; the accumulate kernel (ctrl.hip) cannot afford those ~34 VGPRs across its
// contraction phase.
template <unsigned long long BITS>
FFK_HD double pinned_bits() {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned lo, hi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "n"(static_cast<unsigned>(BITS & 0xffffffffull)));
    asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "n"(static_cast<unsigned>(BITS >> 32)));
    return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
#else
    return __builtin_bit_cast(double, BITS);
#endif
}
#define FFK_PIN(c) (PIN ? pinned_bits<__builtin_bit_cast(unsigned long long, static_cast<double>(c))>() : (c))

template <bool PIN = false>
FFK_HD void sincos_reduced(double r, double* s, double* c) {
    const double z = r*r;
    // sin
    double ps = fma(z, FFK_PIN(1.58969099521155010221e-10), FFK_PIN(-2.50507602534068634195e-08));
    ps = fma(z, ps, FFK_PIN(2.75573137070700676789e-06));
    ps = fma(z, ps, FFK_PIN(-1.98412698298579493134e-04));
    ps = fma(z, ps, FFK_PIN(8.33333333332248946124e-03));
    ps = fma(z, ps, FFK_PIN(-1.66666666666666324348e-01));
    *s = fma(r*z, ps, r);
    // cos
    double pc = fma(z, FFK_PIN(-1.13596475577881948265e-11), FFK_PIN(2.08757232129817482790e-09));
    pc = fma(z, pc, FFK_PIN(-2.75573143513906633035e-07));
    pc = fma(z, pc, FFK_PIN(2.48015872894767294178e-05));
    pc = fma(z, pc, FFK_PIN(-1.38888888888741095749e-03));
    pc = fma(z, pc, FFK_PIN(4.16666666666666019037e-02));
    const double hz = 0.5*z;
    const double w = 1.0 - hz;
    *c = w + (((1.0 - w) - hz) + z*z*pc);
}

// the same kernels cut to the terms that matter for |r| < 2^-5 (z < 2^-10: the dropped terms are
// below 2^-60 relative)
template <bool PIN = false>
FFK_HD void sincos_small(double r, double* s, double* c) {
    const double z = r*r;
    double ps = fma(z, FFK_PIN(2.75573137070700676789e-06), FFK_PIN(-1.98412698298579493134e-04));
    ps = fma(z, ps, FFK_PIN(8.33333333332248946124e-03));
    ps = fma(z, ps, FFK_PIN(-1.66666666666666324348e-01));
    *s = fma(r*z, ps, r);
    double pc = fma(z, FFK_PIN(2.48015872894767294178e-05), FFK_PIN(-1.38888888888741095749e-03));
    pc = fma(z, pc, FFK_PIN(4.16666666666666019037e-02));
    const double hz = 0.5*z;
    const double w = 1.0 - hz;
    *c = w + (((1.0 - w) - hz) + z*z*pc);
}

FFK_HD unsigned low_word(double t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<unsigned>(__double2loint(t));
#else
    unsigned long long bits;
    __builtin_memcpy(&bits, &t, sizeof bits);
    return static_cast<unsigned>(bits);
#endif
}

template <bool PIN = false>
FFK_HD void sincos_pi(double x, double* s, double* c) {
    const double kMagic = FFK_PIN(6755399441055744.0);                     // 1.5 * 2^52
    const double t = fma(x, FFK_PIN(6.36619772367581382433e-01), kMagic);  // x * 2/pi, rounded to integer
    const unsigned q = low_word(t);
    const double k = t - kMagic;
    double r = fma(-k, FFK_PIN(1.57079632679489655800e+00), x);            // pi/2 high
    r = fma(-k, FFK_PIN(6.12323399573676603587e-17), r);                   // pi/2 mid
    r = fma(-k, FFK_PIN(-1.49738490485916983294e-33), r);                  // pi/2 low
    double sr, cr;
    sincos_reduced<PIN>(r, &sr, &cr);
    const double s0 = (q & 1u) ? cr : sr;
    const double c0 = (q & 1u) ? sr : cr;
    *s = (q & 2u) ? -s0 : s0;
    *c = ((q + 1u) & 2u) ? -c0 : c0;
}

#undef FFK_PIN

// Reciprocal to ~1 ulp.  Device: v_rcp_f64 seed + two Newton steps; host: plain division.
FFK_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
#else
    return 1.0/x;
#endif
}

// Reciprocal to ~1e-15 relative (10 ulp): v_rcp_f64 (4.6e-8 on gfx950, measured over 2^20 random
// arguments of either sign and 40 binades) + ONE Newton step.  For the integral entries of the
// accumulate kernels, where the factor 1/x multiplies a sine that is itself good to ~1 ulp and
// 256 segments are summed: two instructions less per entry and frequency.
FFK_HD double rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, y, 1.0);
    return fma(y, e, y);
#else
    return 1.0/x;
#endif
}

// 1/sqrt(x) to ~1 ulp for normal x > 0.  Device: v_rsq_f64 seed + two Newton steps.
FFK_HD double rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x*y, y, 1.0);
    y = fma(y, 0.5*e, y);
    e = fma(-x*y, y, 1.0);
    y = fma(y, 0.5*e, y);
    return y;
#else
    return 1.0/sqrt(x);
#endif
}

// First-order Magnus integral of one (m, n) entry, numeric.py:144-167 with util.cexpm1
// (util.py:165-182):
//     I = (exp(i x dt) - 1)/(i x),  x = omega + dE,  I = dt where x == 0 exactly.
// exp(iy) - 1 = -2 sin^2(y/2) + i sin(y) and sin(y) = 2 sin(y/2) cos(y/2), so with
// s = sin(y/2), c = cos(y/2):  I = (2 s / x) (c + i s).  Cancellation free like the
// reference's form; one sincos instead of two sin.
FFK_HD cplx first_order_integral(double omega, double dE, double dt) {
    const double x = omega + dE;              // same rounding as np.add.outer(E, dE)
    const double y = x*dt;                    // same rounding as int_buf.imag*dt
    double s, c;
    sincos_pi(0.5*y, &s, &c);
    const double q = 2.0*s*rcp(x);
    cplx out = {q*c, q*s};
    if (x == 0.0) {
        out.re = dt;
        out.im = 0.0;
    }
    return out;
}

// The same integral for an off-diagonal entry, given sin/cos of the two half-angles
//     a = fl(omega*dt)/2     (per frequency: the diagonal entry's own argument) and
//     b = fl(dE*dt)/2        (per segment and entry: frequency independent, precomputed),
// through the addition theorems  sin(a+b) = sa cb + ca sb,  cos(a+b) = ca cb - sa sb:
// 4 flops instead of a sincos.  x = fl(omega + dE) (for 1/x and the exact-zero test) is still
// formed exactly like the reference; only the half-angle a+b differs from the reference's
// fl(fl(omega+dE)*dt)/2 by a few ulp of (|omega|+|dE|) dt, worth ~1e-16*dt absolute in I.
// Near a resonance (omega ~ -dE) the sum sa cb + ca sb cancels; below |h| < 2^-5 the direct
// evaluation is used instead (a divergent but rare branch: the band is a few percent of a
// logarithmic grid, for the half of the entries with dE < 0).
FFK_HD cplx first_order_integral_aa(double omega, double dE, double dt, double sa, double ca,
                                    double sb, double cb) {
    const double x = omega + dE;
    const double h = 0.5*(x*dt);
    double s = fma(sa, cb, ca*sb);
    double c = fma(ca, cb, -(sa*sb));
    if (fabs(h) < 0.03125) sincos_small<true>(h, &s, &c);   // no range reduction needed here
    const double q = 2.0*s*rcp_fast(x);
    cplx out = {q*c, q*s};
    if (x == 0.0) {
        out.re = dt;
        out.im = 0.0;
    }
    return out;
}

// The accumulate kernels need E = e^{i omega t_g} I, not I: with h = (omega + dE) dt / 2,
//     I = e^{ih} 2 sin(h)/x      =>      E = e^{i(omega t_g + a)} e^{ib} (2 sin(a + b)/x),
// a = fl(omega dt)/2 per (segment, frequency), b = fl(dE dt)/2 per (segment, entry).  Per frequency
// and segment (PhasedFrequency): psi = e^{i omega t_g} e^{ia} and 2 sin a, 2 cos a; per entry one
// rotation of psi by b (4 flops), 2 sin(a + b) by the addition theorem (2), a reciprocal, 3 products:
// 13 instructions and no select, against 25 for "I, then multiply by the phase" (cos(a + b) is never
// formed, the phase costs nothing extra, the x == 0 limit lives in the rare branch).  Near a
// resonance (|x dt| < 2^-4, which includes x == 0 and zero-length segments) the sine comes from the
// short polynomial -- the sum would cancel -- as in first_order_integral_aa.
// E = psi e^{ib} q with q = 2 sin(a + b)/x REAL: the d = 4 kernel (ctrl_pq.hip) never forms E; it
// folds e^{ib} into the frequency-independent operands and contracts with q (phased_q) alone.
struct PhasedFrequency {
    double om, dt, thr;        // frequency, segment length, 2^-4/dt
    double pr, pi;             // psi = e^{i omega t_g} e^{i a}
    double sa2, ca2;           // 2 sin a, 2 cos a
};
FFK_HD PhasedFrequency phased_frequency(double om, double dt, cplx ph, double sa, double ca) {
    PhasedFrequency f;
    f.om = om;
    f.dt = dt;
    // (a zero-length segment: every entry takes the exact branch, which yields I = 0 also at x == 0)
    f.thr = dt > 0.0 ? 0.0625*rcp(dt) : __builtin_huge_val();
    f.pr = fma(ph.re, ca, -(ph.im*sa));
    f.pi = fma(ph.re, sa, ph.im*ca);
    f.sa2 = sa + sa;
    f.ca2 = ca + ca;
    return f;
}
// the real factor 2 sin(a + b)/x of an entry (x == 0: dt)
FFK_HD double phased_q(const PhasedFrequency& f, double dE, double sb, double cb) {
    const double x = f.om + dE;
    double q;
    if (fabs(x) < f.thr) {
        double s, c;
        sincos_small<true>(0.5*(x*f.dt), &s, &c);     // (the cosine is dead code)
        q = x == 0.0 ? f.dt : 2.0*s*rcp_fast(x);
    } else {
        q = fma(f.sa2, cb, f.ca2*sb)*rcp_fast(x);
    }
    return q;
}
FFK_HD cplx phased_integral_aa(const PhasedFrequency& f, double dE, double sb, double cb) {
    const double er = fma(f.pr, cb, -(f.pi*sb));
    const double ei = fma(f.pr, sb, f.pi*cb);
    const double q = phased_q(f, dE, sb, cb);
    return {q*er, q*ei};
}

// exp(i x) as (cos, sin), util.py:136-162
FFK_HD cplx cexp(double x) {
    cplx out;
    sincos_pi(x, &out.im, &out.re);
    return out;
}

}  // namespace ffk

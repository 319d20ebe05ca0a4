// device_math.h -- scalar device helpers shared by every kernel of librovmpc (gfx950).
//
// Each helper cites the reference statement it restates (paths relative to the reference
// root).  Everything is templated on the arithmetic type T (double | float).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rovmpc.h"

namespace rovmpc {

#define RV_DEV __device__ __forceinline__

RV_DEV double m_sin(double x) { return ::sin(x); }
RV_DEV float  m_sin(float x)  { return ::sinf(x); }
RV_DEV double m_cos(double x) { return ::cos(x); }
RV_DEV float  m_cos(float x)  { return ::cosf(x); }
RV_DEV void m_sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
RV_DEV void m_sincos(float x, float *s, float *c) { ::sincosf(x, s, c); }
RV_DEV double m_sinh(double x) { return ::sinh(x); }
RV_DEV float  m_sinh(float x)  { return ::sinhf(x); }
RV_DEV double m_cosh(double x) { return ::cosh(x); }
RV_DEV float  m_cosh(float x)  { return ::coshf(x); }
RV_DEV double m_tanh(double x) { return ::tanh(x); }
RV_DEV float  m_tanh(float x)  { return ::tanhf(x); }
RV_DEV double m_atanh(double x) { return ::atanh(x); }
RV_DEV float  m_atanh(float x)  { return ::atanhf(x); }
RV_DEV double m_exp(double x) { return ::exp(x); }
RV_DEV float  m_exp(float x)  { return ::expf(x); }
RV_DEV double m_log(double x) { return ::log(x); }
RV_DEV float  m_log(float x)  { return ::logf(x); }
RV_DEV double m_sqrt(double x) { return ::sqrt(x); }
RV_DEV float  m_sqrt(float x)  { return ::sqrtf(x); }
RV_DEV double m_pow(double x, double y) { return ::pow(x, y); }
RV_DEV float  m_pow(float x, float y)  { return ::powf(x, y); }
RV_DEV double m_abs(double x) { return ::fabs(x); }
RV_DEV float  m_abs(float x)  { return ::fabsf(x); }
RV_DEV double m_min(double a, double b) { return ::fmin(a, b); }
RV_DEV float  m_min(float a, float b)  { return ::fminf(a, b); }
RV_DEV double m_max(double a, double b) { return ::fmax(a, b); }
RV_DEV float  m_max(float a, float b)  { return ::fmaxf(a, b); }
RV_DEV bool m_finite(double x) { return ::isfinite(x); }
RV_DEV bool m_finite(float x)  { return ::isfinite(x); }

template <typename T> RV_DEV T m_eps();
template <> RV_DEV double m_eps<double>() { return 2.220446049250313e-16; }
template <> RV_DEV float  m_eps<float>()  { return 1.1920929e-7f; }
template <typename T> RV_DEV T m_nan() { return (T)__builtin_nan(""); }
template <typename T> RV_DEV T m_inf() { return (T)__builtin_inf(); }

// np.clip(x, lo, hi): NaN propagates (fmin/fmax would drop it).
template <typename T> RV_DEV T m_clip(T x, T lo, T hi) {
    return (x != x) ? x : (x < lo ? lo : (x > hi ? hi : x));
}

template <typename T> struct V3 { T x, y, z; };
template <typename T> RV_DEV T dot3(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> RV_DEV V3<T> cross3(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// main_fun.py:18-35 rodrigues_rotation with a unit axis k and precomputed sin/cos:
//   v cos + (k x v) sin + k (k.v)(1 - cos)
template <typename T> RV_DEV V3<T> rodrigues_unit(V3<T> v, V3<T> k, T s, T c) {
    V3<T> kv = cross3(k, v);
    T kd = dot3(k, v) * (T(1) - c);
    return {v.x * c + kv.x * s + k.x * kd, v.y * c + kv.y * s + k.y * kd, v.z * c + kv.z * s + k.z * kd};
}

// Rotation axes of transform_catenary for connection vector rel (main_fun.py:75-89, 102-103):
// theta axis = normalize(xy_projection(rel)) x z (fallbacks [1,0,0] / [0,1,0] below 1e-9),
// gamma axis = rel / |rel| (no epsilon in the reference: |rel| = 0 yields NaN there too).
template <typename T> RV_DEV void theta_gamma_axes(V3<T> rel, V3<T> &th_axis, V3<T> &ga_axis) {
    T nxy = m_sqrt(rel.x * rel.x + rel.y * rel.y);
    T ex, ey;
    if (nxy < T(1e-9)) { ex = T(1); ey = T(0); } else { T inv = T(1) / nxy; ex = rel.x * inv; ey = rel.y * inv; }
    // cross([ex,ey,0],[0,0,1]) = [ey,-ex,0]; its norm is 1 (unit xy) so the second fallback
    // of main_fun.py:86-89 can only trigger for a NaN input.
    th_axis = {ey, -ex, T(0)};
    T inv = T(1) / m_sqrt(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
    ga_axis = {rel.x * inv, rel.y * inv, rel.z * inv};
}

// f(C) of main_fun.py:423, same expression order.
template <typename T> RV_DEV T catenary_f(T C, T l, T L2mH2) {
    T s = m_sinh(T(0.5) * l * C);
    return C * C * L2mH2 - T(4) * s * s;
}

// solve_catenary (main_fun.py:418-431): the root of f on [c_lo, c_hi] that scipy brentq
// returns, NaN exactly when brentq raises (no sign change: f(c_lo) f(c_hi) > 0).
// The root is found by Newton on h(u) = sinh(u) - r u  (u = l C / 2, r = sqrt(L^2-dH^2)/l),
// started at the upper bound u0 of the root given by sinh(u)/u >= 1 + u^2/6 + u^4/120:
// h is convex and increasing right of its minimum, so the iteration descends monotonically
// onto the root and converges quadratically (4-6 iterations for the cable geometries of the
// data set).  *u_out receives l C / 2 so callers can reuse sinh(u) = r u.
template <typename T> RV_DEV T solve_catenary_C(T l, T dH, T L, T c_lo, T c_hi) {
    T L2 = L * L - dH * dH;
    T flo = catenary_f(c_lo, l, L2);
    T fhi = catenary_f(c_hi, l, L2);
    bool ok = !(flo * fhi > T(0)) && m_finite(flo) && m_finite(fhi);
    T r = m_sqrt(L2) / l;
    T u = m_sqrt(m_max(T(60) * (T(-1.0 / 6.0) + m_sqrt(T(1.0 / 36.0) + (r - T(1)) * T(1.0 / 30.0))), T(0)));
    if (!(ok && m_finite(u) && u > T(0))) { u = T(1); ok = false; }
    if (ok) {
        const T tol = T(2) * m_eps<T>();
        for (int it = 0; it < 60; ++it) {
            T sh = m_sinh(u), ch = m_cosh(u);
            T h = sh - r * u;
            T un = u - h / (ch - r);
            if (!(m_finite(un) && un > T(0))) un = u;
            bool done = m_abs(un - u) <= tol * m_abs(un) || m_abs(h) <= tol * sh;
            u = un;
            if (done) break;
        }
    }
    T C = T(2) * u / l;
    if (flo == T(0)) C = c_lo;
    if (fhi == T(0)) C = c_hi;
    return ok ? C : m_nan<T>();
}

// Tension rule of main_fun.py:302-305: T = (w/L) l / (2 sinh(C l / 2)), NaN -> (w/L) l / 2.
template <typename T> RV_DEV T cable_tension(T l, T C, T w_per_len) {
    T Tn = (w_per_len * l) / (T(2) * m_sinh(C * l / T(2)));
    return (Tn != Tn) ? w_per_len * l / T(2) : Tn;
}

// Lowest z (in the "up" sense) of transform_catenary(A, A+rel, Catenary(L), theta, gamma)[3]
// (main_fun.py:38-111 + fully_augmented_catenary.py:21-22), returned relative to A.z.
// The M samples of the theta-rotated catenary are q_j = (t_j B'x, t_j B'y, up*s_j) relative
// to A with s_j = (cosh(C'(l' t_j - x0)) - cosh(C' x0))/C'; the reference then applies
// Rodrigues(-theta) about the theta axis and Rodrigues(+gamma) about the gamma axis to every
// point; only z is needed here, and z_j = m . q_j with m = third row of
// R_gamma(gamma) R_theta(-theta) = rodrigues(r3, theta_axis, +theta), r3 = third row of
// R_gamma -- one 3-vector per node instead of two Rodrigues per point.
template <typename T>
RV_DEV T augmented_lowest_z(V3<T> rel, T theta, T gamma, T L, int M, T up, T c_lo, T c_hi) {
    V3<T> kt, kg;
    theta_gamma_axes(rel, kt, kg);
    T st, ct, sg, cg;
    m_sincos(theta, &st, &ct);
    m_sincos(gamma, &sg, &cg);
    V3<T> Bp = rodrigues_unit(rel, kt, st, ct);                     // main_fun.py:92
    T omc = T(1) - cg;
    V3<T> r3 = {-kg.y * sg + omc * kg.z * kg.x, kg.x * sg + omc * kg.z * kg.y, cg + omc * kg.z * kg.z};
    V3<T> m = rodrigues_unit(r3, kt, st, ct);
    T lp = m_sqrt(Bp.x * Bp.x + Bp.y * Bp.y);
    T dHp = up * Bp.z;
    T Cp = solve_catenary_C(lp, dHp, L, c_lo, c_hi);
    T best;
    if (Cp == Cp) {
        T x0 = T(0.5) * lp - m_atanh(dHp / L) / Cp;
        T ch0 = m_cosh(Cp * x0);
        T invC = T(1) / Cp;
        T hx = m.x * Bp.x + m.y * Bp.y;          // horizontal part of m . q_j is t_j * hx
        T mz = m.z * up;
        best = m_inf<T>();
        T denom = T(M - 1);
        for (int j = 0; j < M; ++j) {
            T t = T(j) / denom;
            T s = (m_cosh(Cp * (lp * t - x0)) - ch0) * invC;
            T z = up * (t * hx + mz * s);
            best = (z != z) ? z : (z < best ? z : best);     // np.min propagates NaN
        }
    } else {
        // catenary_fn(...)[3] is None -> straight segment [A, B'] (main_fun.py:67-69)
        T zb = up * dot3(m, Bp);
        best = (zb != zb) ? zb : (zb < T(0) ? zb : T(0));
    }
    return up * best;
}

// order-preserving double <-> int64 map (signed compare of keys == IEEE compare of values)
RV_DEV long long ordered_key(double v) {
    long long b = __double_as_longlong(v);
    return b ^ ((b >> 63) & 0x7FFFFFFFFFFFFFFFLL);
}
RV_DEV double ordered_val(long long k) {
    long long b = k ^ ((k >> 63) & 0x7FFFFFFFFFFFFFFFLL);
    return __longlong_as_double(b);
}

}  // namespace rovmpc

// device_math.h -- scalar device helpers shared by every kernel of librovmpc (gfx950).
//
// Each helper cites the reference statement it restates (paths relative to the reference
// root).  Everything is templated on the arithmetic type T (double | float).
#pragma once
#if !defined(__HIPCC_RTC__)   // hiprtc pre-includes its own HIP runtime declarations
#include <hip/hip_runtime.h>
#endif
#if defined(ROVMPC_JIT_BUILD)
#include "rovmpc.h"
#else
#include <stdint.h>
#include "../../include/rovmpc.h"
#endif

namespace rovmpc {

#define RV_DEV __device__ __forceinline__

template <bool B> struct BoolC { static constexpr bool value = B; };   // compile-time flag for generic lambdas
template <bool B, typename X, typename Y> struct RvCond { typedef X type; };        // (std::conditional: hiprtc has no <type_traits>)
template <typename X, typename Y> struct RvCond<false, X, Y> { typedef Y type; };

// fp64 sin/cos.  The device library's double-precision sin/cos carry their argument
// reduction and polynomial in double-double arithmetic (~120 VALU instructions per call, a
// Payne-Hanek branch on top); the rollout's sequential phase issues up to ten of them per
// horizon step, which made it the kernel's critical path.  For |x| < 2^26 the two-FMA
// Cody-Waite reduction r = x - k (pi/2)_hi - k (pi/2)_lo is exact to < 1 ulp of r and the
// fdlibm minimax kernels (k_sin.c / k_cos.c, degree 13 / 14) give sin and cos of r to < 1 ulp:
// ~35 instructions for the pair.  Larger arguments take the library path (a branch no rollout
// on physical data ever enters).
// Out-of-line so the six call sites of the sequential phase do not each inline ~150
// instructions of Payne-Hanek reduction they never execute.
__device__ __attribute__((noinline)) double2 slow_sincos_f64(double x) { double s, c; ::sincos(x, &s, &c); return make_double2(s, c); }

// The 15 fp64 literals of the fast path.  gfx950 VALU instructions cannot carry a 64-bit
// literal, so each use of a coefficient costs two s_mov_b32 (or, kept live across a loop, SGPR
// pairs that spill to VGPR lanes).  pin() launders them into VGPRs once per kernel phase.
struct TrigK {
    double inv_pio2, pio2_hi, pio2_lo, S1, S2, S3, S4, S5, S6, C1, C2, C3, C4, C5, C6;
};
RV_DEV double pin_vgpr(double v) { asm volatile("" : "+v"(v)); return v; }
RV_DEV TrigK trig_constants(bool pin) {
    TrigK k = {6.36619772367581382433e-01, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
               -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
               2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
               4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
               -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11};
    if (pin) {
        double *p = &k.inv_pio2;
        #pragma unroll
        for (int i = 0; i < 15; ++i) p[i] = pin_vgpr(p[i]);
    }
    return k;
}

RV_DEV void fast_sincos_k(double x, const TrigK &K, double *s, double *c) {
    const bool big = !(::fabs(x) < 67108864.0);
    const double k = ::rint(x * K.inv_pio2);
    double r = ::fma(-k, K.pio2_hi, x);
    r = ::fma(-k, K.pio2_lo, r);
    const double z = r * r;
    const double ps = K.S1 + z * (K.S2 + z * (K.S3 + z * (K.S4 + z * (K.S5 + z * K.S6))));
    const double sr = ::fma(z * r, ps, r);
    const double pc = K.C1 + z * (K.C2 + z * (K.C3 + z * (K.C4 + z * (K.C5 + z * K.C6))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * z * pc);
    const int q = (int)k;
    const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    // signs: bit 1 of q (sine), bit 1 of q + 1 (cosine) into the sign bit
    double rs = __hiloint2double(__double2hiint(ss) ^ (int)(((unsigned)q & 2u) << 30), __double2loint(ss));
    double rc = __hiloint2double(__double2hiint(cc) ^ (int)(((unsigned)(q + 1) & 2u) << 30), __double2loint(cc));
    // large arguments (and NaN): the library's reduction, behind ONE wave-uniform branch -- a per-lane branch costs two
    // exec-mask round trips on every call, taken or not
    if (__builtin_amdgcn_ballot_w64(big) != 0) { const double2 r = slow_sincos_f64(x); if (big) { rs = r.x; rc = r.y; } }
    *s = rs; *c = rc;
}
RV_DEV void fast_sincos_f64(double x, double *s, double *c) { fast_sincos_k(x, trig_constants(false), s, c); }

// sin alone: k = rint(x / pi), r = x - k pi in [-pi/2, pi/2] (two FMAs), odd polynomial to r^17 interpolating
// (sin r / r - 1) / r^2 at the Chebyshev nodes of [0, (pi/2)^2] (fitted with 60-digit arithmetic; 3.7e-17 relative with the
// coefficients rounded to double -- the Taylor polynomial needs r^23 for the same), sign (-1)^k.  ~19 instructions instead
// of ~38 for the sincos pair; < 1 ulp against a 40-digit reference for |x| < 1e6.
struct SinK { double inv_pi, pi_hi, pi_lo, c[8]; };
RV_DEV SinK sin_constants(bool pin) {
    SinK k = {0.31830988618379067154, 3.14159265358979311600e+00, 1.22464679914735317723e-16,
              {-0.16666666666666666, 0.008333333333333316, -0.00019841269841254885, 2.7557319219139174e-06, -2.5052107613933993e-08, 1.6058977114025027e-10, -7.643963964341869e-13, 2.7313658477694193e-15}};
    if (pin) {
        double *p = &k.inv_pi;
        #pragma unroll
        for (int i = 0; i < 11; ++i) p[i] = pin_vgpr(p[i]);
    }
    return k;
}
constexpr double TRIG_FAST_LIMIT = 67108864.0;     // |x| below this: two-term Cody-Waite reduction is exact enough
template <bool CHECKED = true>
RV_DEV double fast_sin_k(double x, const SinK &K) {
    const bool big = CHECKED && !(::fabs(x) < TRIG_FAST_LIMIT);
    const double k = ::rint(x * K.inv_pi);
    double r = ::fma(-k, K.pi_hi, x);
    r = ::fma(-k, K.pi_lo, r);
    const double z = r * r;
    // Estrin's scheme: three dependent levels behind z instead of Horner's seven (a wave issues a dependent instruction
    // every ~9.5 cycles, an independent one every ~5: the two extra multiplications are free)
    const double z2 = z * z, z4 = z2 * z2;
    const double p01 = ::fma(K.c[1], z, K.c[0]), p23 = ::fma(K.c[3], z, K.c[2]), p45 = ::fma(K.c[5], z, K.c[4]), p67 = ::fma(K.c[7], z, K.c[6]);
    const double p = ::fma(::fma(p67, z2, p45), z4, ::fma(p23, z2, p01));
    const double v = ::fma(r * z, p, r);
    // sign (-1)^k: bit 0 of k into the sign bit (three instructions; the select form takes five)
    double res = __hiloint2double(__double2hiint(v) ^ (int)((unsigned)(int)k << 31), __double2loint(v));
    if (CHECKED && __builtin_amdgcn_ballot_w64(big) != 0) { const double sl = slow_sincos_f64(x).x; if (big) res = sl; }
    return res;
}
// fp32 (BASELINE config 3).  The device library's sinf / sincosf inline a Payne-Hanek reduction beside every call and
// evaluate both polynomials for either result: ~40 instructions and six mask operations on the hot path, twice the fp64
// chain's own sine.  Same construction as above in single precision: k = rint(x / pi), r = x - k pi by two FMAs (exact
// product, |k| < 2^14 below the limit), odd polynomial to r^9 fitted on [-pi/2, pi/2] (1.9 ulp measured against fp64 over
// 2e5 points), sign (-1)^k -- 13 instructions, no branch; the pair reduces by pi/2 and uses the fdlibm single-precision
// kernels (k_sinf.c / k_cosf.c).  Larger arguments (and NaN) take the library, out of line.
constexpr float TRIGF_FAST_LIMIT = 32768.0f;
__device__ __attribute__((noinline)) float slow_sinf_f32(float x) { return ::sinf(x); }
__device__ __attribute__((noinline)) float2 slow_sincos_f32(float x) { float s, c; ::sincosf(x, &s, &c); return make_float2(s, c); }
template <bool CHECKED = true>
RV_DEV float fast_sinf_k(float x) {
    const bool big = CHECKED && !(::fabsf(x) < TRIGF_FAST_LIMIT);
    const float k = ::rintf(x * 0.318309886f);
    float r = ::fmaf(-k, 3.14159274f, x);
    r = ::fmaf(-k, -8.74227766e-8f, r);
    const float z = r * r;
    float p = ::fmaf(z, 2.6324394e-06f, -0.00019822021f);
    p = ::fmaf(z, p, 0.008333237f);
    p = ::fmaf(z, p, -0.16666666f);          // (Horner: Estrin's two levels for these four terms cost C3 0.45 us -- measured)
    const float v = ::fmaf(r * z, p, r);
    float res = ((int)k & 1) ? -v : v;
    if (CHECKED && __builtin_amdgcn_ballot_w64(big) != 0) { const float sl = slow_sinf_f32(x); if (big) res = sl; }
    return res;
}
RV_DEV void fast_sincosf_k(float x, float *s, float *c) {
    const bool big = !(::fabsf(x) < TRIGF_FAST_LIMIT);
    const float k = ::rintf(x * 0.636619772f);
    float r = ::fmaf(-k, 1.57079637f, x);
    r = ::fmaf(-k, -4.37113883e-8f, r);
    const float z = r * r;
    float ps = ::fmaf(z, 2.7183114e-06f, -0.00019839335f);
    ps = ::fmaf(z, ps, 0.0083333294f);
    ps = ::fmaf(z, ps, -0.16666667f);
    const float sr = ::fmaf(r * z, ps, r);
    float pc = ::fmaf(z, 2.4390449e-05f, -0.0013886764f);
    pc = ::fmaf(z, pc, 0.041666623f);
    const float cr = ::fmaf(z * z, pc, ::fmaf(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
    float rs = (q & 2) ? -ss : ss, rc = ((q + 1) & 2) ? -cc : cc;
    if (__builtin_amdgcn_ballot_w64(big) != 0) { const float2 r = slow_sincos_f32(x); if (big) { rs = r.x; rc = r.y; } }
    *s = rs; *c = rc;
}

// Trig context: fp64 carries the pinned constants, fp32 the kernels above (constants are 32-bit literals).
template <typename T> struct Trig;
template <> struct Trig<double> {
    TrigK K;
    SinK S;
    RV_DEV explicit Trig(bool pin) : K(trig_constants(pin)), S(sin_constants(pin)) {}
    RV_DEV double sin(double x) const { return fast_sin_k<true>(x, S); }
    // caller guarantees |x| < TRIG_FAST_LIMIT (and x is not NaN): no large-argument branch on the chain
    RV_DEV double sin_bounded(double x) const { return fast_sin_k<false>(x, S); }
    static RV_DEV bool bounded(double b) { return b < TRIG_FAST_LIMIT; }
    RV_DEV void sincos(double x, double *s, double *c) const { fast_sincos_k(x, K, s, c); }
};
template <> struct Trig<float> {
    RV_DEV explicit Trig(bool) {}
    RV_DEV float sin(float x) const { return fast_sinf_k<true>(x); }
    RV_DEV float sin_bounded(float x) const { return fast_sinf_k<false>(x); }
    static RV_DEV bool bounded(float b) { return b < TRIGF_FAST_LIMIT; }
    RV_DEV void sincos(float x, float *s, float *c) const { fast_sincosf_k(x, s, c); }
};
// sine alone (expressions of loaded models): the one-polynomial kernel, ~22 instructions against ~38 for the pair;
// above TRIG_FAST_LIMIT it takes the library's reduction like the pair does
RV_DEV double m_sin(double x) { return fast_sin_k<true>(x, sin_constants(false)); }
RV_DEV float  m_sin(float x)  { return fast_sinf_k<true>(x); }
RV_DEV double m_cos(double x) { double s, c; fast_sincos_f64(x, &s, &c); return c; }
RV_DEV float  m_cos(float x)  { float s, c; fast_sincosf_k(x, &s, &c); return c; }
RV_DEV void m_sincos(double x, double *s, double *c) { fast_sincos_f64(x, s, c); }
RV_DEV void m_sincos(float x, float *s, float *c) { fast_sincosf_k(x, s, c); }
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
RV_DEV float  m_sqrt(float x)  { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32: 1 ulp, no fix-up sequence
RV_DEV double m_pow(double x, double y) { return ::pow(x, y); }
RV_DEV float  m_pow(float x, float y)  { return ::powf(x, y); }
RV_DEV double m_fma(double a, double b, double c) { return ::fma(a, b, c); }
RV_DEV float  m_fma(float a, float b, float c)  { return ::fmaf(a, b, c); }
RV_DEV double m_abs(double x) { return ::fabs(x); }
RV_DEV float  m_abs(float x)  { return ::fabsf(x); }
RV_DEV double m_min(double a, double b) { return ::fmin(a, b); }
RV_DEV float  m_min(float a, float b)  { return ::fminf(a, b); }
RV_DEV double m_max(double a, double b) { return ::fmax(a, b); }
RV_DEV float  m_max(float a, float b)  { return ::fmaxf(a, b); }
// minNum without the canonicalising v_max the compiler puts in front of fmin's operands (one per operand that is not known to
// be quiet): v_min_f64 / v_min_f32 in the kernels' IEEE mode quiet a signalling NaN themselves.  Inner loops only.
RV_DEV double m_min_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RV_DEV float  m_min_raw(float a, float b)  { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RV_DEV bool m_finite(double x) { return ::isfinite(x); }
RV_DEV bool m_finite(float x)  { return ::isfinite(x); }

// 1/x for finite positive x away from the denormals: v_rcp_f64 (~26 bits) + two Newton steps, ~1 ulp, 5
// instructions instead of the ~18 of an IEEE division.  Not for x that may be 0 or inf (0 * inf = NaN).
RV_DEV double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = ::fma(r, ::fma(-x, r, 1.0), r);
    return ::fma(r, ::fma(-x, r, 1.0), r);
}
RV_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32: 1 ulp
// a / b on the geometry chains: in fp64 the nine-instruction quotient of m_divq below; in fp32 (parity rule: same arg-min or
// |dJ|/J < 1e-4) one v_rcp_f32 and a multiplication instead of the ten-instruction division sequence
RV_DEV double m_divq(double a, double b);
RV_DEV double m_div(double a, double b) { return m_divq(a, b); }      // (1 ulp, special operands as IEEE: see m_divq)
// a / b inside a loaded model's expression (hiprtc path).  The compiler's IEEE sequence is eleven instructions, two of them
// v_div_scale for operands whose quotient leaves the exponent range; an expression over scaled features of order one does
// not need those: reciprocal + two Newton steps + one residual correction (<= 1 ulp), and v_div_fixup for the special
// operands -- x / 0 is still +-inf, 0 / 0 and NaN still NaN, x / inf still 0, as NumPy has them.
RV_DEV double m_divq(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = ::fma(r, ::fma(-b, r, 1.0), r);
    r = ::fma(r, ::fma(-b, r, 1.0), r);
    double q = a * r;
    q = ::fma(::fma(-b, q, a), r, q);
    return __builtin_amdgcn_div_fixup(q, b, a);
}
RV_DEV float m_divq(float a, float b) { return a / b; }
// The same for `/` inside a LOADED model's expression (hiprtc path, interpreter), whose operands are whatever the user's
// features are (generation 2 is unscaled): the reciprocal form is exact to 1 ulp while neither 1 / b nor the quotient leaves the
// exponent range, which |a| < 2^500 and 2^-500 < |b| < 2^500 guarantee; anything else (and NaN) takes the IEEE sequence.
RV_DEV double m_divx(double a, double b) {
    const double ab = ::fabs(b);
    if (!(ab > 0x1p-500 && ab < 0x1p500 && ::fabs(a) < 0x1p500)) return a / b;
    return m_divq(a, b);
}
RV_DEV float m_divx(float a, float b) { return a / b; }
// sqrt of a squared norm on the integrating wave (|v|^2, |v x u|^2: zero or comfortably inside the exponent range): v_rsq_f64
// with one Goldschmidt step and one residual correction (1 ulp), without the operand scaling and the second correction of
// the compiler's sequence -- eleven instructions instead of seventeen.  0 and inf map to themselves, NaN and negatives to NaN.
RV_DEV double m_sqrtq(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = ::fma(-h, g, 0.5);
    g = ::fma(g, r, g); h = ::fma(h, r, h);
    g = ::fma(::fma(-g, g, x), h, g);
    return (x == 0.0 || x == __builtin_inf()) ? x : g;
}
RV_DEV float m_sqrtq(float x) { return m_sqrt(x); }
RV_DEV float  m_div(float a, float b)  { return a * __builtin_amdgcn_rcpf(b); }

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
// The same about a horizontal unit axis (kx, ky, 0) -- the theta axis of main_fun.py:75-89 always is: the terms with
// k.z dropped in the source, because the compiler may not drop `0 * x` itself (x could be inf or NaN) and leaves four
// multiplications by a literal zero on the dependent chain.  Same value for finite operands.
template <typename T> RV_DEV V3<T> rodrigues_flat(V3<T> v, T kx, T ky, T s, T c) {
    const T kd = (kx * v.x + ky * v.y) * (T(1) - c);
    return {v.x * c + (ky * v.z) * s + kx * kd, v.y * c - (kx * v.z) * s + ky * kd, v.z * c + (kx * v.y - ky * v.x) * s};
}

// Rotation axes of transform_catenary for connection vector rel (main_fun.py:75-89, 102-103):
// theta axis = normalize(xy_projection(rel)) x z (fallbacks [1,0,0] / [0,1,0] below 1e-9),
// gamma axis = rel / |rel| (no epsilon in the reference: |rel| = 0 yields NaN there too).
template <typename T> RV_DEV void theta_gamma_axes(V3<T> rel, V3<T> &th_axis, V3<T> &ga_axis) {
    T nxy = m_sqrtq(rel.x * rel.x + rel.y * rel.y);
    T ex, ey;
    if (nxy < T(1e-9)) { ex = T(1); ey = T(0); } else { T inv = m_div(T(1), nxy); ex = rel.x * inv; ey = rel.y * inv; }
    // cross([ex,ey,0],[0,0,1]) = [ey,-ex,0]; its norm is 1 (unit xy) so the second fallback
    // of main_fun.py:86-89 can only trigger for a NaN input.
    th_axis = {ey, -ex, T(0)};
    T inv = m_div(T(1), m_sqrtq(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z));
    ga_axis = {rel.x * inv, rel.y * inv, rel.z * inv};
}

// f(C) of main_fun.py:423, same expression order.
template <typename T> RV_DEV T catenary_f(T C, T l, T L2mH2) {
    T s = m_sinh(T(0.5) * l * C);
    return C * C * L2mH2 - T(4) * s * s;
}

// sinh(x)/x - 1 and cosh(x) - 1 for 0 <= x < 0.5 without cancellation (8 series terms,
// truncation < 1e-19 relative at x = 0.5).
template <typename T> RV_DEV T sinhc_m1_small(T x2) {
    return x2 * T(1.0 / 6) * (T(1) + x2 * T(1.0 / 20) * (T(1) + x2 * T(1.0 / 42) * (T(1) + x2 * T(1.0 / 72) *
           (T(1) + x2 * T(1.0 / 110) * (T(1) + x2 * T(1.0 / 156) * (T(1) + x2 * T(1.0 / 210) * (T(1) + x2 * T(1.0 / 272))))))));
}
template <typename T> RV_DEV T cosh_m1_small(T x2) {
    return x2 * T(0.5) * (T(1) + x2 * T(1.0 / 12) * (T(1) + x2 * T(1.0 / 30) * (T(1) + x2 * T(1.0 / 56) *
           (T(1) + x2 * T(1.0 / 90) * (T(1) + x2 * T(1.0 / 132) * (T(1) + x2 * T(1.0 / 182) * (T(1) + x2 * T(1.0 / 240))))))));
}

// sinh(x) for x >= 0: series below 0.5, (e^x - e^-x)/2 above (relative error ~1 ulp either way).
template <typename T> RV_DEV T sinh_pos(T x) {
    if (x < T(0.5)) return x * (T(1) + sinhc_m1_small(x * x));
    const T e = m_exp(x);
    return T(0.5) * (e - T(1) / e);
}

// Result of the catenary-parameter solve: C (NaN if none), u = l C / 2, r = sqrt(L^2-dH^2)/l -- so that sinh(u) = r u at
// the root (reused by the tension rule and the shape samples) -- and e = exp(u) (carried out of the iteration: the shape
// samples and the warm start of a nearby system need it, and the iteration has it for a handful of instructions).
template <typename T> struct CatRoot { T C, u, r, e, sq; };   // sq = sqrt(L^2 - dH^2)

// e^d for |d| < 2^-5 (fp64: Taylor to d^7, truncation < 2e-17 relative; fp32: to d^4, < 3e-10): how exp(u) is carried from
// one Halley iterate to the next, u' = u + d, instead of being evaluated again.
RV_DEV double exp_increment(double d) {
    double p = 1.0 / 5040;
    p = ::fma(p, d, 1.0 / 720); p = ::fma(p, d, 1.0 / 120); p = ::fma(p, d, 1.0 / 24);
    p = ::fma(p, d, 1.0 / 6); p = ::fma(p, d, 0.5); p = ::fma(p, d, 1.0);
    return ::fma(p, d, 1.0);
}
// the same for a converged iterate, |d| < 2^-15: to d^3 (< 2e-20 relative; the only literal is 1/6)
RV_DEV double exp_increment_tiny(double d) { return ::fma(::fma(::fma(d, 1.0 / 6, 0.5), d, 1.0), d, 1.0); }
RV_DEV float exp_increment_tiny(float d) { return ::fmaf(::fmaf(d, 0.5f, 1.0f), d, 1.0f); }
RV_DEV float exp_increment(float d) {
    float p = ::fmaf(d, 1.0f / 24, 1.0f / 6);
    p = ::fmaf(p, d, 0.5f); p = ::fmaf(p, d, 1.0f);
    return ::fmaf(p, d, 1.0f);
}

// solve_catenary (main_fun.py:418-431): the root C* of f(C) = C^2 (L^2 - dH^2) - 4 sinh^2(l C / 2)
// that scipy brentq returns on [c_lo, c_hi], NaN when brentq raises.  brentq raises exactly
// when f(c_lo) f(c_hi) > 0; f is positive below its single positive root and negative above
// (and negative everywhere for a taut cable, L^2 - dH^2 <= l^2), so that is the same as
// "no root, or the root outside [c_lo, c_hi]" -- decided here from the root itself instead of
// two more sinh evaluations (the two statements differ only when the root is within an ulp of
// a bracket end).
//
// The root comes from Halley's iteration on h(u) = sinh(u) - r u, u = l C / 2,
// r = sqrt(L^2 - dH^2) / l:
//  * start: the root of 1 + u^2/6 + u^4/120 = r (an upper bound, exact to O(u^6)), tightened
//    by two steps of u <- log(2 r u) when r > 8;
//  * near-taut cables (u < 0.5) use the cancellation-free forms h = u (S(u) - (r-1)),
//    h' = (cosh u - 1) - (r - 1) with r - 1 = (L2 - l^2) / (l (sqrt(L2) + l));
//  * two iterations are unrolled (cubic convergence from a start within 1 %: the second step is < 1e-6 relative for
//    r < 4.5); systems whose last step was still > 1e-6 keep iterating, so the result never depends on the unroll count;
//  * exp(u) is evaluated once: an iterate that moved by |d| < 2^-5 takes e^u e^d with the short polynomial of d (round 3:
//    an exp is ~30 instructions, the increment 8; the rollout's per-node geometry ran seven exps, now two).
// Warm start (WARM; u_warm, e_warm = exp(u_warm)): the root of a NEARBY system -- the rollout solves the same cable at
// the same node twice, for the end point and for the end point turned by theta about a horizontal axis, a few
// parts in a thousand apart -- accepted when h'(u_warm) = cosh u_warm - r > 0 (Newton-type steps from there land
// above the root and descend).  Two iterations then reach 1e-14 (measured over the benchmark's geometry); a system
// without a usable warm start takes the series bound and the tail loop runs the extra iterations it needs.
template <typename T, bool WARM>
RV_DEV CatRoot<T> solve_catenary_root_impl(T l, T dH, T L, T c_lo, T c_hi, T u_warm, T e_warm) {
    const T L2 = L * L - dH * dH;
    const T sq = m_sqrtq(L2);
    // 1 / l once, for r here and C at the end (l = 0 gives NaN, l = inf gives 0: either way not a valid system below, as
    // with the quotients themselves)
    const T il = fast_rcp(l);
    T r = sq * il;
    // r - 1 without cancellation; only the near-taut forms and the cold start need it (a warm solve away from u < 0.5 never does)
    auto r_minus_1 = [&]() { return m_div(L2 - l * l, l * (sq + l)); };
    T u, e = T(-1);                  // e = exp(u) of the current iterate, or < 0: not known
    bool ok;
    bool warm = false;
    if (WARM) {
        const T chw = T(0.5) * (e_warm + fast_rcp(e_warm));          // (e_warm > 0 whenever u_warm > 0)
        warm = u_warm > T(0) && chw > r && L2 > l * l && m_finite(r);
    }
    if (warm) {
        u = u_warm; e = e_warm;
        ok = true;
    } else {
        const T rm1 = r_minus_1();
        // root of y/6 + y^2/120 = r - 1 in y = u^2 (an upper bound, exact to O(u^6)), then one Newton step on the series with
        // its y^3/5040 term: the start is within 0.4 % of the root at r = 3.7 (2 % at r = 8) instead of 3 % (8 %), so that two
        // Halley iterations reach the last bit where three were needed, and the second one already moves by |d| < 2^-5
        T y = m_max(T(60) * (T(-1.0 / 6.0) + m_sqrtq(T(1.0 / 36.0) + rm1 * T(1.0 / 30.0))), T(0));
        y = y - (y * y * y * T(1.0 / 5040)) * fast_rcp(T(1.0 / 6) + y * (T(1.0 / 60) + y * T(1.0 / 1680)));
        u = m_sqrtq(y);
        ok = rm1 > T(0) && m_finite(r) && m_finite(u) && u > T(0);
        if (!ok) { u = T(1); r = T(2); }
        if (r > T(8)) {
            T ul = m_log(T(2) * r * u);
            ul = m_log(T(2) * r * ul) + T(0.05);
            if (ul > T(0) && ul < u) u = ul;
        }
    }
    bool moving = true;               // the last step changed u by more than 1e-6 relative
    auto halley = [&](auto LAST) {       // LAST: the last unrolled iteration -- a converged step carries exp(u) by the short form
        T h, hp, sh;
        if (u < T(0.5)) {
            const T rm1 = ok ? r_minus_1() : T(1);
            const T u2 = u * u;
            const T S = sinhc_m1_small(u2);
            sh = u * (T(1) + S);
            h = u * (S - rm1);
            hp = cosh_m1_small(u2) - rm1;
            e = T(-1);
        } else {
            if (e < T(0)) e = m_exp(u);                              // 1 <= e < e^(c_hi L / 2): finite, positive
            const T ei = fast_rcp(e);
            sh = T(0.5) * (e - ei);
            h = sh - r * u;
            hp = T(0.5) * (e + ei) - r;
        }
        T un = u - (T(2) * h * hp) * fast_rcp(T(2) * hp * hp - h * sh);   // a zero or infinite denominator gives NaN: caught below
        if (!(m_finite(un) && un > T(0))) un = u;
        const T d = un - u;
        moving = m_abs(d) > T(1e-6) * un;
        if (LAST.value && !moving && m_abs(d) < T(3e-5)) e = e > T(0) ? e * exp_increment_tiny(d) : T(-1);
        else e = (e > T(0) && m_abs(d) < T(0.03125)) ? e * exp_increment(d) : T(-1);
        u = un;
    };
    halley(BoolC<false>{}); halley(BoolC<true>{});
    for (int it = 0; it < 40 && moving && ok; ++it) halley(BoolC<false>{});
    if (e < T(0)) e = m_exp(u);                                      // (near-taut roots, u < 0.5: the series path carries no exp)
    const T C = (T(2) * u) * il;
    const bool in = ok && C >= c_lo && C <= c_hi;
    return {in ? C : m_nan<T>(), u, r, e, sq};
}

template <typename T> RV_DEV CatRoot<T> solve_catenary_root(T l, T dH, T L, T c_lo, T c_hi) {
    return solve_catenary_root_impl<T, false>(l, dH, L, c_lo, c_hi, T(0), T(1));
}
template <typename T> RV_DEV CatRoot<T> solve_catenary_root_warm(T l, T dH, T L, T c_lo, T c_hi, T u_warm, T e_warm) {
    return solve_catenary_root_impl<T, true>(l, dH, L, c_lo, c_hi, u_warm, e_warm);
}

template <typename T> RV_DEV T solve_catenary_C(T l, T dH, T L, T c_lo, T c_hi) {
    return solve_catenary_root(l, dH, L, c_lo, c_hi).C;
}

// Tension rule of main_fun.py:302-305: T = (w/L) l / (2 sinh(C l / 2)), NaN -> (w/L) l / 2.
// At the root sinh(C l / 2) = sinh(u) = r u, which the solve already holds.
template <typename T> RV_DEV T cable_tension(T l, CatRoot<T> c, T w_per_len) {
    return (c.C == c.C) ? (w_per_len * l) * fast_rcp(T(2) * c.r * c.u) : w_per_len * l / T(2);     // (valid root: r u = sinh u > 0, finite)
}

// Lowest z (in the "up" sense) of transform_catenary(A, A+rel, Catenary(L), theta, gamma)[3]
// (main_fun.py:38-111 + fully_augmented_catenary.py:21-22), relative to A.z, in two halves
// around the catenary solve of the theta-rotated end point B'.
// The M samples of the theta-rotated catenary are q_j = (t_j B'x, t_j B'y, up*s_j) relative
// to A with s_j = (cosh(C'(l' t_j - x0)) - cosh(C' x0))/C'; the reference then applies
// Rodrigues(-theta) about the theta axis and Rodrigues(+gamma) about the gamma axis to every
// point; only z is needed here, and z_j = m . q_j with m = third row of
// R_gamma(gamma) R_theta(-theta) = rodrigues(r3, theta_axis, +theta), r3 = third row of
// R_gamma -- one 3-vector per node instead of two Rodrigues per point.
template <typename T> struct AugShape { V3<T> Bp, m; T lp, dHp; };

// (st, ct) = sincos(theta), (sg, cg) = sincos(gamma)
template <typename T>
RV_DEV AugShape<T> augmented_prepare(V3<T> rel, V3<T> kt, V3<T> kg, T st, T ct, T sg, T cg, T up) {
    AugShape<T> a;
    a.Bp = rodrigues_flat(rel, kt.x, kt.y, st, ct);                  // main_fun.py:92 (kt.z == 0 by construction, :75-89)
    const T omc = T(1) - cg;
    const V3<T> r3 = {-kg.y * sg + omc * kg.z * kg.x, kg.x * sg + omc * kg.z * kg.y, cg + omc * kg.z * kg.z};
    a.m = rodrigues_flat(r3, kt.x, kt.y, st, ct);
    a.lp = m_sqrtq(a.Bp.x * a.Bp.x + a.Bp.y * a.Bp.y);
    a.dHp = up * a.Bp.z;
    return a;
}

// The cosh samples are equally spaced in their argument a + j d, a = atanh(dH'/L) - u',
// d = 2u'/(M-1), so e^{a+jd} and e^{-(a+jd)} advance by one multiplication each
// (e^a = sqrt((L+dH')/(L-dH')) e^{-u'}, e^{u'} comes with the root): one exp per node instead of M cosh + atanh.
// Sample j (reference: t_j = j / (M - 1), s_j = (cosh(C'(l' t_j - x0)) - cosh(C' x0)) / C', z_j = up (t_j hx + mz C' s_j)) is
//   z_j = j a + b (E_j + 1/E_j) + c0,   a = up hx / (M - 1),  b = m.z / (2 C'),  c0 = -b (E_0 + 1/E_0):
// two products, two sums, one FMA and the minimum per sample (round 3; the literal form took twelve instructions).  np.min
// propagates NaN: a NaN in any sample is a NaN in the last one (E_j, 1/E_j and the running j a keep it), except that
// 0 * inf at j = 0 needs hx and mz themselves finite -- both are checked once behind the loop instead of once per sample.
// inv_Mm1 = 1 / (M - 1), from the host.
template <typename T>
RV_DEV T augmented_finish(const AugShape<T> &a, CatRoot<T> c, T L, int M, T inv_Mm1, T up) {
    T best;
    if (c.C == c.C) {
        // a valid root means L^2 - dH'^2 > l'^2 > 0: every quantity inverted below is finite and positive
        // e^{a} = sqrt((L + dH') / (L - dH')) e^{-u'} = (L + dH') / (sqrt(L^2 - dH'^2) e^{u'}): root and exp come with the solve
        T E = (L + a.dHp) * fast_rcp(c.sq * c.e);
        T Ei = fast_rcp(E);
        const T Ed = m_exp(T(2) * c.u * inv_Mm1), Edi = fast_rcp(Ed);
        const T hx = up * (a.m.x * a.Bp.x + a.m.y * a.Bp.y);   // horizontal part of m . q_j is t_j * hx (up = +-1 folded in:
        const T mz = a.m.z * fast_rcp(c.C);                    //  the reference's mz carries up as well, up * up = 1)
        const T sa = hx * inv_Mm1, sb = T(0.5) * mz;
        T lin = -(sb * (E + Ei));                        // c0; + a per sample
        T z = T(0);                                      // sample 0 is the anchor itself
        best = T(0);
        int j = 1;
        if constexpr (sizeof(T) == 4) {
            // fp32: two consecutive samples per trip as one packed pair -- (E_j, E_j+1) advance by Ed^2 with one v_pk_mul_f32,
            // the sums and the FMA are v_pk_add_f32 / v_pk_fma_f32: 7 instructions per two samples instead of 12
            typedef float f2 __attribute__((ext_vector_type(2)));
            const float Ed2 = Ed * Ed, Edi2 = Edi * Edi;
            f2 Ev = {E * Ed, E * Ed2}, Eiv = {Ei * Edi, Ei * Edi2};
            f2 linv = {lin + sa, lin + (sa + sa)};
            const f2 stepE = {Ed2, Ed2}, stepI = {Edi2, Edi2}, stepL = {sa + sa, sa + sa}, sbv = {sb, sb};
            for (; j + 1 < M; j += 2) {
                const f2 zv = __builtin_elementwise_fma(sbv, Ev + Eiv, linv);
                best = m_min_raw(m_min_raw(best, zv.x), zv.y);
                z = zv.y;
                Ev *= stepE; Eiv *= stepI; linv += stepL;
            }
            // (an odd sample is left when M - 1 is odd: the scalar loop below takes it from the pair's state)
            E = Ev.x * Edi; Ei = Eiv.x * Ed; lin = linv.x - sa;
        }
#pragma unroll 3
        for (; j < M; ++j) {
            E *= Ed; Ei *= Edi;
            lin += sa;
            z = m_fma(sb, E + Ei, lin);
            best = m_min_raw(best, z);                   // minNum: a NaN operand is dropped; see above
        }
        const T chk = z + ((hx - hx) + (mz - mz));       // NaN iff a sample was (or hx / mz is not finite)
        if (chk != chk) best = m_nan<T>();
    } else {
        // catenary_fn(...)[3] is None -> straight segment [A, B'] (main_fun.py:67-69)
        const T zb = up * dot3(a.m, a.Bp);
        best = (zb != zb) ? zb : (zb < T(0) ? zb : T(0));
    }
    return up * best;
}

// The four lanes of a quad (lanes 4q .. 4q+3) each receive all four lanes' values:
// out[j] = v of lane 4 (lane / 4) + j.  One v_mov_b32_dpp quad_perm per 32-bit half and source
// lane -- a VALU move, no LDS round trip.  (Measured on MI355X for the role exchange of the
// sequential phase: ds_bpermute ~150 cycles per exchange, v_permlane16/32_swap rows slower still.)
template <int J> RV_DEV unsigned quad_bcast_u32(unsigned v) {
    constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);          // quad_perm:[J,J,J,J]
    // (mov_dpp: every lane of a full quad has a source, so the destination needs no prior value -- update_dpp with an
    // `old` operand costs a v_mov per use to materialise it)
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xf, 0xf, false);
}
RV_DEV void quad4(double v, double (&out)[4]) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    out[0] = __hiloint2double((int)quad_bcast_u32<0>(hi), (int)quad_bcast_u32<0>(lo));
    out[1] = __hiloint2double((int)quad_bcast_u32<1>(hi), (int)quad_bcast_u32<1>(lo));
    out[2] = __hiloint2double((int)quad_bcast_u32<2>(hi), (int)quad_bcast_u32<2>(lo));
    out[3] = __hiloint2double((int)quad_bcast_u32<3>(hi), (int)quad_bcast_u32<3>(lo));
}
RV_DEV void quad4(float v, float (&out)[4]) {
    const unsigned u = __float_as_uint(v);
    out[0] = __uint_as_float(quad_bcast_u32<0>(u)); out[1] = __uint_as_float(quad_bcast_u32<1>(u));
    out[2] = __uint_as_float(quad_bcast_u32<2>(u)); out[3] = __uint_as_float(quad_bcast_u32<3>(u));
}

// Lane i of a 16-lane row reads lane i + SH of the same row (lanes without a source keep their
// own value): v_mov_b32_dpp row_shl.  Used for the arg-min over <= 16 candidates of a workgroup.
template <int SH> RV_DEV unsigned row_shl_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x100 + SH, 0xf, 0xf, false);
}
template <int SH> RV_DEV double row_shl(double v) {
    return __hiloint2double((int)row_shl_u32<SH>((unsigned)__double2hiint(v)), (int)row_shl_u32<SH>((unsigned)__double2loint(v)));
}
template <int SH> RV_DEV long long row_shl(long long v) {
    const unsigned lo = row_shl_u32<SH>((unsigned)v), hi = row_shl_u32<SH>((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// features_dd (main_fun.py:842-843): v_surge = V . unit_rel, v_sway = |V x unit_rel|
template <typename T> RV_DEV void dd_surge_sway(T vx, T vy, T vz, T ux, T uy, T uz, T &sway, T &surge) {
    surge = vx * ux + vy * uy + vz * uz;
    const T cx = vy * uz - vz * uy, cy = vz * ux - vx * uz, cz = vx * uy - vy * ux;
    sway = m_sqrtq(cx * cx + cy * cy + cz * cz);
}

// ---- counter-based normals (the proposal law of MPC.step with device sampling; util_kernels.h states the law) ---------
RV_DEV void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned *out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the four standard normals of block j: Philox4x32-10(counter (j lo, j hi, step lo, step hi), key (seed lo, seed hi)),
// u_i = (x_i + 0.5) 2^-32, two Box-Muller pairs.  One routine for the stand-alone sampler and for the rollout kernel that
// draws its candidates itself, so that both produce the same bits.
RV_DEV void philox_normal4(unsigned long long seed, unsigned long long step, long long j, double (&z)[4]) {
    unsigned x[4];
    philox4x32_10((unsigned)j, (unsigned)((unsigned long long)j >> 32), (unsigned)step, (unsigned)(step >> 32),
                  (unsigned)seed, (unsigned)(seed >> 32), x);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double u1 = ((double)x[2 * h] + 0.5) * 2.3283064365386963e-10, u2 = ((double)x[2 * h + 1] + 0.5) * 2.3283064365386963e-10;
        const double r = ::sqrt(-2.0 * ::log(u1));
        double sn, cs;
        m_sincos(6.283185307179586 * u2, &sn, &cs);
        z[2 * h] = r * cs; z[2 * h + 1] = r * sn;
    }
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

//------------------------------------------------------------------------------
///  @file prelude.hpp
///  @brief The device helpers every generated kernel starts with: the window checks, the
///  shared-reciprocal division in both precisions, pow(x, 1.5).
///
///  Division contract.  The reference's cpu_context divides with the compiler's IEEE `/`
///  (arithmetic.hpp:3508).  A work item divides many numerators by few denominators (680
///  divisions, 82 denominators in the RK4 kernel), so the reciprocal is refined once per
///  denominator and shared.
///
///  fp64.  hipcc lowers `/` to: v_div_scale of both operands, r = rcp(d) refined by Newton steps,
///  q = n*r, a residual step, v_div_fmas (un-scale), v_div_fixup (special operands).  The default
///  mode spends three fused multiply-adds per quotient — the SAME instructions the compiler's
///  sequence executes when v_div_scale leaves both operands unscaled, hence the same bits.  The
///  hardware scales when the denominator, its reciprocal, the quotient or the residual would leave
///  the normal range; every pass therefore checks, per lane:
///    * every denominator: finite, non-zero, |d| inside [2^-500, 2^500];
///    * every value the pass stores and every gather argument: finite (an overflowing or
///      non-finite numerator turns into a NaN here, where IEEE division gives an infinity);
///    * every stored value computed from a quotient: not zero (without v_div_fixup a quotient
///      with numerator -0 is +0; a zero keeps its place through every later operation, so it
///      can only be seen in a stored zero — or in a division by it, which the first check sees —
///      or by a pow node, which maps -0 and +0 to different infinities or zeros when its exponent is
///      an odd integer: the base of every pow node that comes from a quotient is checked for zero as
///      well, unless its exponent is a constant that is not an odd integer; codegen.hpp, GFIR_POW);
///    * `checked` mode only: every numerator is zero or |n| >= 2^-450, so that neither the
///      residual nor the quotient can be subnormal.
///  What the default mode leaves unchecked in fp64, and `checked` closes at 1.5 vector instructions
///  per numerator: a non-zero numerator below 2^-969, or a quotient below the normal range, may
///  differ from IEEE division in the last bit.
///
///  fp32 (round 2).  Quotients are rounded from an fp64 product, float(double(n)*r) with r the
///  refined fp64 reciprocal of d: the IEEE fp32 quotient for EVERY finite n and finite non-zero d,
///  subnormal and overflowing results and the sign of a zero included (the separation argument is
///  with gf_div below).  The only check left is that every denominator is finite and non-zero.
///
///  A lane that fails a check REDOES THE PASS with the compiler's division (`<name>_ieee`, a
///  function of its own that the hot path never enters on the benchmark workloads) and raises a
///  status bit (informational; bit 0: window/finite, bit 1: stored zero — fp64 only).
//------------------------------------------------------------------------------
#ifndef gfhip_prelude_hpp
#define gfhip_prelude_hpp

#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>
#include <vector>

#include "gfir_item.hpp"
#include "options.hpp"

namespace gfhip {

//  v_div_fixup only acts on special operands.  With the checks above (finite non-zero
//  denominators, finite results, no stored zero that came from a quotient) dropping it changes
//  nothing that can be observed, except through atan2, which tells -0 from +0 (and pow with an odd
//  integer exponent, whose base joins the zero check instead, codegen.hpp): the fixup (680
//  of 7700 VALU instructions in the RK4 step, 8 % of its time) is kept only for items that
//  contain an atan2 node, or on request (GFHIP_DIV_FIXUP=1).
inline bool division_fixup(const item &it, const codegen_options &opt) {
    if (opt.division_fixup >= 0) return opt.division_fixup == 1;
    for (auto &c : it.code) {
        if (c.op == GFIR_ATAN2) return true;
    }
    return false;
}

//  Complex base types (graph_type COMPLEX_FLOAT / COMPLEX_DOUBLE).  The reference's kernels use
//  std::complex with the host compiler's operators (cpu_context.hpp:428-584), i.e. whatever
//  __muldc3/__divdc3 the JIT links; no fixture pins them ("parity unpinned").  Here:
//  the textbook product, Smith's quotient (libgcc's classic __divdc3 without its NaN recovery),
//  every operation unfused; elementary functions through their real parts' formulas.
inline void emit_complex(std::ostringstream &s, const bool f64) {
    const std::string x = f64 ? "" : "f";
    s << R"(
struct gf_complex {
    base re, im;
    __device__ __forceinline__ gf_complex() : re(0), im(0) {}
    __device__ __forceinline__ gf_complex(const base r, const base i) : re(r), im(i) {}
};
__device__ __forceinline__ gf_complex operator+(const gf_complex a, const gf_complex b) { return gf_complex(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ gf_complex operator-(const gf_complex a, const gf_complex b) { return gf_complex(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ gf_complex operator*(const gf_complex a, const gf_complex b) {
    return gf_complex(a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re);
}
__device__ __forceinline__ gf_complex operator/(const gf_complex a, const gf_complex b) {
    if (__builtin_fabs)" << x << R"((b.re) < __builtin_fabs)" << x << R"((b.im)) {
        const base ratio = b.re/b.im, denom = b.re*ratio + b.im;
        return gf_complex((a.re*ratio + a.im)/denom, (a.im*ratio - a.re)/denom);
    }
    const base ratio = b.im/b.re, denom = b.im*ratio + b.re;
    return gf_complex((a.im*ratio + a.re)/denom, (a.im - a.re*ratio)/denom);
}
__device__ __forceinline__ bool operator==(const gf_complex a, const gf_complex b) { return a.re == b.re && a.im == b.im; }
__device__ __forceinline__ base gf_real(const gf_complex a) { return a.re; }
__device__ __forceinline__ gf_complex gf_fma(const gf_complex a, const gf_complex b, const gf_complex c) { return a*b + c; }   // arithmetic.hpp:5118-5121
__device__ __forceinline__ gf_complex gf_exp(const gf_complex a) {
    const base e = exp)" << x << R"((a.re);
    return gf_complex(e*cos)" << x << R"((a.im), e*sin)" << x << R"((a.im));
}
__device__ __forceinline__ gf_complex gf_log(const gf_complex a) {
    return gf_complex(log)" << x << R"((hypot)" << x << R"((a.re, a.im)), atan2)" << x << R"((a.im, a.re));
}
__device__ __forceinline__ gf_complex gf_sqrt(const gf_complex a) {
    if (a.re == 0 && a.im == 0) return gf_complex(0, a.im);
    const base m = hypot)" << x << R"((a.re, a.im);
    if (a.re >= 0) {
        const base t = __builtin_sqrt)" << x << R"(((m + a.re)*static_cast<base> (0.5));
        return gf_complex(t, a.im/(t + t));
    }
    const base t = __builtin_sqrt)" << x << R"(((m - a.re)*static_cast<base> (0.5));
    return gf_complex(__builtin_fabs)" << x << R"((a.im)/(t + t), a.im < 0 ? -t : t);
}
__device__ __forceinline__ gf_complex gf_pow(const gf_complex a, const gf_complex b) {
    if (a.re == 0 && a.im == 0) return (b.re == 0 && b.im == 0) ? gf_complex(1, 0) : gf_complex(0, 0);
    return gf_exp(b*gf_log(a));
}
__device__ __forceinline__ gf_complex gf_sin(const gf_complex a) {
    return gf_complex(sin)" << x << R"((a.re)*cosh)" << x << R"((a.im), cos)" << x << R"((a.re)*sinh)" << x << R"((a.im));
}
__device__ __forceinline__ gf_complex gf_cos(const gf_complex a) {
    return gf_complex(cos)" << x << R"((a.re)*cosh)" << x << R"((a.im), -sin)" << x << R"((a.re)*sinh)" << x << R"((a.im));
}
// atan(z) = (i/2) log((i + z)/(i - z)); the reference emits atan(right/left) for complex types (trigonometry.hpp:718-722)
__device__ __forceinline__ gf_complex gf_atan2(const gf_complex r, const gf_complex l) {
    const gf_complex z = r/l, i(0, 1);
    const gf_complex w = gf_log((i + z)/(i - z));
    return gf_complex(-w.im*static_cast<base> (0.5), w.re*static_cast<base> (0.5));
}
__device__ __forceinline__ gf_complex gf_nan_to_zero(const gf_complex a) {        // cpu_context.hpp:530-537
    return gf_complex(a.re != a.re ? static_cast<base> (0) : a.re, a.im != a.im ? static_cast<base> (0) : a.im);
}
__device__ __forceinline__ gf_complex gf_from_base(const base a) { return gf_complex(a, 0); }
)";
}

//  erfi for complex arguments (erfi_node, math.hpp:1440; the reference calls special::erfi,
//  special_functions.hpp:1583, a port of the Faddeeva package).  Here: erfi(z) = -i erf(iz) with
//  erf(u) = 1 - exp(-u^2) w(iu) and the Faddeeva function w by Weideman's rational approximation
//  (J. A. C. Weideman, SIAM J. Numer. Anal. 31 (1994) 1497: w(z) ~ 2 p(Z)/(L - iz)^2 +
//  (1/sqrt(pi))/(L - iz), Z = (L + iz)/(L - iz), p of degree N - 1 with Fourier coefficients of
//  exp(-t^2)(L^2 + t^2), L = sqrt(N/sqrt 2); lower half plane: w(z) = 2 exp(-z^2) - w(-z)).
//  N = 48 meets the reference's own test of its erfi (graph_tests/erfi_test.cpp:20-83 on
//  graph_tests/test_erfi.nc: |1 - test/gold| <= 2e-14): tests/test_oracle.py, tests/test_gpu_generic.py.
//  Evaluated in double whatever the item's base type.
inline void weideman_coefficients(const int n, double &length, std::vector<double> &a) {
    const int m = 2*n, m2 = 2*m;
    const long double pi = 3.141592653589793238462643383279502884L;
    const long double l = std::sqrt(static_cast<long double> (n)/std::sqrt(2.0L));
    length = static_cast<double> (l);
    std::vector<long double> f(m2, 0.0L);                   // f[0] = 0, f[j] for k = j - m, j = 1 .. 4n - 1
    for (int j = 1; j < m2; j++) {
        const long double t = l*std::tan((j - m)*pi/m/2.0L);
        f[j] = std::exp(-t*t)*(l*l + t*t);
    }
    a.assign(n, 0.0);
    for (int k = 1; k <= n; k++) {                          // real part of the DFT of fftshift(f), / 4n
        long double sum = 0.0L;
        for (int j = 0; j < m2; j++) {
            sum += f[(j + m2/2)%m2]*std::cos(2.0L*pi*k*j/m2);
        }
        a[k - 1] = static_cast<double> (sum/m2);            // coefficient of Z^(k - 1)
    }
}

inline void emit_erfi(std::ostringstream &s) {
    double length;
    std::vector<double> a;
    weideman_coefficients(48, length, a);
    char buf[64];
    s << "__device__ static const double gf_weideman[48] = {";
    for (size_t k = 0; k < a.size(); k++) {
        std::snprintf(buf, sizeof(buf), "%a", a[k]);
        s << (k ? ", " : "") << buf;
    }
    std::snprintf(buf, sizeof(buf), "%a", length);
    s << "};\n#define GF_WEIDEMAN_L " << buf << "\n";
    s << R"(
struct gf_z { double re, im; };
__device__ __forceinline__ gf_z gf_zmul(const gf_z a, const gf_z b) { return gf_z{a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re}; }
__device__ __forceinline__ gf_z gf_zdiv(const gf_z a, const gf_z b) {
    const double d = b.re*b.re + b.im*b.im;
    return gf_z{(a.re*b.re + a.im*b.im)/d, (a.im*b.re - a.re*b.im)/d};
}
__device__ __forceinline__ gf_z gf_zexp(const gf_z a) { const double e = exp(a.re); return gf_z{e*cos(a.im), e*sin(a.im)}; }
// w(z) for Im z >= 0
__device__ inline gf_z gf_faddeeva_upper(const gf_z z) {
    const gf_z down{GF_WEIDEMAN_L + z.im, -z.re};           // L - iz
    const gf_z up{GF_WEIDEMAN_L - z.im, z.re};              // L + iz
    const gf_z big = gf_zdiv(up, down);
    gf_z p{gf_weideman[47], 0.0};
    for (int k = 46; k >= 0; k--) {
        p = gf_zmul(p, big);
        p.re += gf_weideman[k];
    }
    const gf_z first = gf_zdiv(gf_z{2.0*p.re, 2.0*p.im}, gf_zmul(down, down));
    const gf_z second = gf_zdiv(gf_z{0x1.20dd750429b6dp-1, 0.0}, down);       // 1/sqrt(pi)
    return gf_z{first.re + second.re, first.im + second.im};
}
__device__ inline gf_z gf_faddeeva(const gf_z z) {
    if (z.im >= 0.0) return gf_faddeeva_upper(z);
    const gf_z square = gf_zmul(z, z);
    const gf_z e = gf_zexp(gf_z{-square.re, -square.im});
    const gf_z w = gf_faddeeva_upper(gf_z{-z.re, -z.im});
    return gf_z{2.0*e.re - w.re, 2.0*e.im - w.im};
}
// The branches special::erf_complex takes before its general formula (special_functions.hpp:1495-1517),
// as behaviour: on the imaginary axis i erf(Im z); on the real axis the REAL value exp(x^2) Im w(x),
// saturated beyond x^2 = 720 (Im Z(zeta) = sqrt(pi) exp(-zeta^2) then keeps its relative accuracy in
// dispersion.hpp:289-297); Re(z^2) < -750 gives -+i.
__device__ inline gf_complex gf_erfi(const gf_complex a) {
    const gf_z z{static_cast<double> (a.re), static_cast<double> (a.im)};
    if (z.re == 0.0) return gf_complex(a.re, static_cast<base> (erf(z.im)));
    if (z.im == 0.0) {
        const double on_axis = z.re*z.re > 720.0 ? copysign(__DBL_MAX__, z.re)
                                                 : exp(z.re*z.re)*gf_faddeeva_upper(gf_z{z.re, 0.0}).im;
        return gf_complex(static_cast<base> (on_axis), a.im);
    }
    if (isinf(z.re) && isinf(z.im)) return gf_complex(static_cast<base> (0), static_cast<base> (-0.0));
    if ((z.re + z.im)*(z.re - z.im) < -750.0) return gf_complex(static_cast<base> (0), static_cast<base> (z.im <= 0.0 ? -1.0 : 1.0));
// Small arguments: 1 - exp(z^2) w(-z) cancels (Im erfi would keep only ~1e-16/|z| of relative accuracy).  The
// regions are those of special::erf_complex for u = iz = (-Im z) + i(Re z) (special_functions.hpp:1534-1553):
// |Im z| < 0.08 and |Re z| < 0.01: the Maclaurin series erfi(z) = 2/sqrt(pi) sum z^(2k+1)/(k! (2k+1)), k <= 6;
// else |Im z| < 0.005 and |2 Re z Im z| < 0.005: the expansion of erf(x + iy) about the imaginary axis,
//   erf(iy) + (2/sqrt pi) exp(y^2) [x (1 - x^2 (1 + 2y^2)/3 + x^4 (3 + 12y^2 + 4y^4)/30) - i x^2 y (1 - x^2 (3 + 2y^2)/6)],
//   erf(iy) = i exp(y^2) Im w(y).
    if (fabs(z.im) < 8.0e-2) {
        if (fabs(z.re) < 1.0e-2) {
            const gf_z s = gf_zmul(z, z);
            gf_z p{0x1.f9a326f9b89b7p-14, 0.0};                                  // 2/sqrt(pi)/(6! 13)
            const double c[6] = {0x1.c02db40040b85p-11, 0x1.565bcd0e6a53fp-8, 0x1.b82ce31288b51p-6, 0x1.ce2f21a042be2p-4,
                                 0x1.812746b0379e7p-2, 0x1.20dd750429b6dp+0};
            for (int k = 0; k < 6; k++) {
                p = gf_zmul(p, s);
                p.re += c[k];
            }
            const gf_z v = gf_zmul(z, p);
            return gf_complex(static_cast<base> (v.re), static_cast<base> (v.im));
        }
        if (fabs(z.im) < 5.0e-3 && fabs(2.0*z.re*z.im) < 5.0e-3) {
            const double x = -z.im, y = z.re, x2 = x*x, y2 = y*y, e = exp(y2);
            const double wim = gf_faddeeva_upper(gf_z{y, 0.0}).im;
            const double erf_re = e*x*(0x1.20dd750429b6dp+0 - x2*(0x1.812746b0379e7p-2 + 0x1.812746b0379e7p-1*y2)
                                       + x2*x2*(0x1.ce2f21a042be2p-4 + y2*(0x1.ce2f21a042be2p-2 + 0x1.341f6bc02c7ecp-3*y2)));
            const double erf_im = e*(wim - x2*y*(0x1.20dd750429b6dp+0 - x2*(0x1.20dd750429b6dp-1 + 0x1.812746b0379e7p-2*y2)));
            return gf_complex(static_cast<base> (erf_im), static_cast<base> (-erf_re));     // -i erf(iz)
        }
    }
    const gf_z e = gf_zexp(gf_zmul(z, z));                                       // exp(-u^2), u = iz
    const gf_z w = gf_faddeeva(gf_z{-z.re, -z.im});                              // w(iu) = w(-z)
    const gf_z erf{1.0 - (e.re*w.re - e.im*w.im), -(e.re*w.im + e.im*w.re)};    // erf(iz)
    return gf_complex(static_cast<base> (erf.im), static_cast<base> (-erf.re)); // -i erf(iz)
}
)";
}

//  What items with complex values, SAFE_MATH guards or a random state call by name (the plain
//  real-valued hot path keeps the builtins in its text).
inline void emit_generic(std::ostringstream &s, const item &it, const bool f64) {
    const std::string x = f64 ? "" : "f";
    s << R"(
__device__ __forceinline__ base gf_real(const base a) { return a; }
__device__ __forceinline__ base gf_fma(const base a, const base b, const base c) { return __builtin_fma)" << x << R"((a, b, c); }
__device__ __forceinline__ base gf_sqrt(const base a) { return __builtin_sqrt)" << x << R"((a); }
__device__ __forceinline__ base gf_exp(const base a) { return exp)" << x << R"((a); }
__device__ __forceinline__ base gf_log(const base a) { return log)" << x << R"((a); }
__device__ __forceinline__ base gf_pow(const base a, const base b) { return pow)" << x << R"((a, b); }
__device__ __forceinline__ base gf_sin(const base a) { return sin)" << x << R"((a); }
__device__ __forceinline__ base gf_cos(const base a) { return cos)" << x << R"((a); }
__device__ __forceinline__ base gf_atan2(const base r, const base l) { return atan2)" << x << R"((r, l); }
__device__ __forceinline__ base gf_nan_to_zero(const base a) { return a != a ? static_cast<base> (0) : a; }          // cpu_context.hpp:538-541
)";
    if (!it.is_complex()) s << "__device__ __forceinline__ base gf_from_base(const base a) { return a; }\n";
    if (it.has_random()) {
//  random_state_node::mt_state and random_node::compile_random (random.hpp:44-52, :318-339):
//  MT19937 advanced one word per draw.
        s << R"(
struct gf_mt_state { unsigned int array[624]; unsigned short index; };
__device__ inline unsigned int gf_random(gf_mt_state &state) {
    const unsigned short k = state.index;
    unsigned short j = (k + 1)%624;
    unsigned int x = (state.array[k] & 0x80000000u) | (state.array[j] & 0x7fffffffu);
    unsigned int xa = x >> 1;
    if (x & 1u) xa ^= 0x9908b0dfu;
    j = (k + 397)%624;
    x = state.array[j]^xa;
    state.array[k] = x;
    state.index = (k + 1)%624;
    unsigned int y = x^(x >> 11);
    y = y^((y << 7) & 0x9d2c5680u);
    y = y^((y << 15) & 0xefc60000u);
    return y^(y >> 18);
}
)";
    }
}

inline void emit_prelude(std::ostringstream &s, const item &it, const codegen_options &opt, const size_t pack_count) {
    const bool f64 = it.base_is_f64();
    const char *base = f64 ? "double" : "float";
    s << "// Generated by graph_framework_amd (GFIR -> gfx950).  Work item \"" << it.name << "\": "
      << it.code.size() << " nodes, " << it.tables.size() << " tables in " << pack_count << " packs.\n";
//  hipRTC predefines the runtime declarations; its include search does not always reach the
//  ROCm headers (a stand-alone process on a box whose /opt/rocm hipRTC has no header path).
    s << "#if !defined(__HIPCC_RTC__)\n#include <hip/hip_runtime.h>\n#endif\n";
    s << "typedef " << base << " base;\n";
    if (it.is_complex()) {
        emit_complex(s, f64);
        for (auto &c : it.code) {
            if (c.op == GFIR_ERFI) {
                emit_erfi(s);
                break;
            }
        }
        s << "typedef gf_complex real;\n";
    } else {
        s << "typedef base real;\n";
    }
    if (it.is_complex() || it.safe_math() || it.has_random()) emit_generic(s, it, f64);
    if (it.is_complex()) return;
    const bool fixup = division_fixup(it, opt);
//  Window check of the denominators: |d| is tracked through an fp32 image that is monotonic in
//  |d| — |d| itself for float, the high dword of a double read as a float (sign, 11 exponent
//  bits, 20 mantissa bits) — so that one v_maximum3_f32 / v_minimum3_f32 (gfx950, abs modifiers
//  free) folds TWO denominators into the running extreme: 1 VALU instruction per denominator
//  instead of 3 with fp64 fmax/fmin.  Both are the IEEE-754-2019 NaN-propagating forms: a
//  high dword that reads as a float NaN (|d| >= 2^1017, infinity, NaN) poisons the accumulator
//  and fails the final comparison, like any other value outside the window.
//  Numerators (`checked` mode): key = (bits << 1) - 1 on the (high) dword drops the sign and
//  maps a zero to the largest key, so an unsigned minimum finds the smallest NON-ZERO magnitude
//  (v_lshl_add_u32 + half a v_min3_u32 per numerator).
    s << R"(
__device__ __forceinline__ float gf_magnitude(const float d) { return __builtin_fabsf(d); }
__device__ __forceinline__ float gf_magnitude(const double d) {
    return __builtin_fabsf(__builtin_bit_cast(float, static_cast<unsigned int> (__builtin_bit_cast(unsigned long long, d) >> 32)));
}
__device__ __forceinline__ unsigned int gf_numerator_key(const float n) { return (__builtin_bit_cast(unsigned int, n) << 1) - 1u; }
__device__ __forceinline__ unsigned int gf_numerator_key(const double n) {
    return (static_cast<unsigned int> (__builtin_bit_cast(unsigned long long, n) >> 32) << 1) - 1u;
}
)";
    const bool fast = opt.division == division_mode::fast;
    if (!f64) {
        s << R"(
// fp32 quotients through fp64.  The exact quotient n/d of two floats that is not itself a rounding
// boundary of the fp32 format (a float, or the midpoint of two neighbouring floats: at most 25
// significant bits M) is at least 1/(M*D) > 2^-49 of its own size away from it (n 2^k - M d is a
// non-zero integer for 24-bit n, d), subnormal and overflowing results included; a double within
// 2^-51 of the quotient therefore rounds to the IEEE fp32 quotient, sign of a zero included.
// r = rcp(d) in fp64 refined by one cubic step (e = 1 - d r, r' = r (1 + e + e^2): |e| <= 2^-22 from
// v_rcp_f64 leaves e^3 < 2^-66, the two roundings 2^-52; shared per denominator: 5 instructions),
// q = float(double(n)*r) (3 per quotient) — and no condition on the numerator, the quotient or a
// stored zero is left to check: only that every denominator is finite and non-zero (the step turns
// rcp(0) = inf and rcp(inf) = 0 into NaN).
__device__ __forceinline__ double gf_rcp(const float d) {
    const double wide = d;
    const double r = __builtin_amdgcn_rcp(wide);
    const double e = __builtin_fma(-wide, r, 1.0);
    return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
// sqrtf(x) for a finite x >= 2^-96: the compiler's sqrtf scales arguments below 2^-96 by 2^32, takes v_sqrt_f32
// (1 ulp), tests its two neighbours with one fma each, un-scales and keeps a zero or an infinity (16 vector
// instructions); at and above 2^-96 the scale is 1 and the selection takes the root, so these nine give the same
// bits (the correctly rounded root).  codegen.hpp tracks the argument's extremes; a negative x is NaN here as there.
__device__ __forceinline__ float gf_sqrt_window(const float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float above = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float e_below = __builtin_fmaf(-below, s, x);
    const float e_above = __builtin_fmaf(-above, s, x);
    const float t = e_below <= 0.0f ? below : s;
    return e_above > 0.0f ? above : t;
}
__device__ __forceinline__ float gf_div(const float n, const float, const double r) {
    return static_cast<float> (static_cast<double> (n)*r);
}
)";
    }
    if (f64) {
        s << (fixup ? "#define GF_FIXUP(q, d, n) __builtin_amdgcn_div_fixup(q, d, n)\n"
                    : "#define GF_FIXUP(q, d, n) (q)\n");
        s << "#define GF_FAST_DIVISION " << (fast ? 1 : 0) << "\n";
        s << R"(
// fp64 division as hipcc lowers it: r = rcp(d) refined by two Newton steps (shared per
// denominator), q = n*r, e = fma(-d, q, n), q' = fma(e, r, q).
__device__ __forceinline__ double gf_rcp(const double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double gf_div(const double n, const double d, const double r) {
#if GF_FAST_DIVISION
    return n*r;             // GFHIP_DIVISION=fast: within ~1.5 ulp of the quotient, NOT the IEEE quotient
#else
    const double q = n*r;
    const double e = __builtin_fma(-d, q, n);
    return GF_FIXUP(__builtin_fma(e, r, q), d, n);
#endif
}
// sqrt(x) for x inside the window the pass checks (finite, |x| in [2^-500, 2^500]; codegen.hpp tracks the argument
// with the denominators): hipcc lowers sqrt to  scale x by 2^256 if x < 2^-767 -> y = rsq(x) -> g = x*y, h = y/2 ->
// one coupled Newton step on (g, h) -> two residual steps -> un-scale -> keep x if it is 0 or inf  (18 vector
// instructions).  Inside the window the scale is 0 and the final selection takes the computed root, so the nine
// instructions below are exactly the ones that sequence executes on such an x: the same bits.  A negative x gives
// NaN here (rsq) as there.
__device__ __forceinline__ double gf_sqrt_window(const double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x*y;
    const double h0 = y*0.5;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    return __builtin_fma(d1, h1, g2);
}
// pow(x, 1.5) = x*sqrt(x) with the rounding error of the square root carried into the
// product (s + t ~ sqrt(x) to ~100 bits), i.e. rounded once from the exact value almost
// always, as glibc's pow is; ocml's pow is within 1 ulp only and costs ~10x more.
__device__ __forceinline__ double gf_pow_three_halves(const double x) {
    const double s = __builtin_sqrt(x);
    const double t = __builtin_fma(-s, s, x)*(0.5*__builtin_amdgcn_rcp(s));
    const double p = x*s;
    const double c = __builtin_fma(x, s, -p) + x*t;
    return (x > 0.0 && x < __builtin_inf()) ? p + c : pow(x, 1.5);
}
// The same for an x inside the checked window and positive (a negative x: NaN, as pow(x, 1.5) gives): no selection,
// no call.
__device__ __forceinline__ double gf_pow_three_halves_window(const double x) {
    const double s = gf_sqrt_window(x);
    const double t = __builtin_fma(-s, s, x)*(0.5*__builtin_amdgcn_rcp(s));
    const double p = x*s;
    const double c = __builtin_fma(x, s, -p) + x*t;
    return p + c;
}
)";
    }
}

}  // namespace gfhip

#endif /* gfhip_prelude_hpp */

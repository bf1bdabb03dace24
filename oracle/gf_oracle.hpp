// ---------------------------------------------------------------------------
// gf_oracle.hpp — CPU ORACLE (test infrastructure, NOT product code).
//
// A plain C++ restatement of the reference's algorithm for the hot path:
// EFIT bicubic-spline equilibrium -> cold-plasma dispersion D -> ray ODE right
// hand side (symbolic df() in the reference, forward-mode duals here) ->
// Newton `loss_kernel` and RK4 `solver_kernel`, plus the xkorc push.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// use this.  The product path (graph_framework_amd/csrc) never includes it.
//
// Every function cites the reference file:line it follows (paths relative to
// the reference checkout, /root/reference).  The reference differentiates
// symbolically and then algebraically reduces the DAG; forward-mode dual
// numbers evaluate the same derivatives of the same expressions, so results
// agree up to floating-point re-association only.
//
// Pinning: see oracle/README.md (efit_gold.nc known-answer test + SURVEY.md
// §8(c) probe values of the reference run).
// ---------------------------------------------------------------------------
#ifndef GF_ORACLE_HPP
#define GF_ORACLE_HPP

#include <cmath>
#include <cstddef>
#include <vector>
#include <algorithm>

namespace gfo {

// ---------------------------------------------------------------------------
// Forward-mode dual number with N tangents.  Plays the role of
// graph::leaf_node::df (node.hpp:364 ff.; e.g. multiply_node::df
// arithmetic.hpp:1720 ff., divide_node::df :2769 ff., sqrt_node::df
// math.hpp:26 ff.).
// ---------------------------------------------------------------------------
template<typename T, int N>
struct Dual {
    T v;
    T d[N];

    Dual() : v(0) { for (int i = 0; i < N; i++) d[i] = 0; }
    Dual(const T c) : v(c) { for (int i = 0; i < N; i++) d[i] = 0; }
    static Dual seed(const T value, const int index) {
        Dual r(value);
        r.d[index] = 1;
        return r;
    }
};

template<typename T, int N>
inline Dual<T, N> operator+(const Dual<T, N> &a, const Dual<T, N> &b) {
    Dual<T, N> r; r.v = a.v + b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i];
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator-(const Dual<T, N> &a, const Dual<T, N> &b) {
    Dual<T, N> r; r.v = a.v - b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i];
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator-(const Dual<T, N> &a) {
    Dual<T, N> r; r.v = -a.v;
    for (int i = 0; i < N; i++) r.d[i] = -a.d[i];
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator*(const Dual<T, N> &a, const Dual<T, N> &b) {
    Dual<T, N> r; r.v = a.v*b.v;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i]*b.v + a.v*b.d[i];
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator/(const Dual<T, N> &a, const Dual<T, N> &b) {
    Dual<T, N> r; r.v = a.v/b.v;
//  d(a/b) = da/b - a*db/(b*b)   (divide_node::df form)
    for (int i = 0; i < N; i++) r.d[i] = a.d[i]/b.v - a.v*b.d[i]/(b.v*b.v);
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator+(const Dual<T, N> &a, const T b) { Dual<T, N> r = a; r.v = a.v + b; return r; }
template<typename T, int N>
inline Dual<T, N> operator+(const T a, const Dual<T, N> &b) { Dual<T, N> r = b; r.v = a + b.v; return r; }
template<typename T, int N>
inline Dual<T, N> operator-(const Dual<T, N> &a, const T b) { Dual<T, N> r = a; r.v = a.v - b; return r; }
template<typename T, int N>
inline Dual<T, N> operator-(const T a, const Dual<T, N> &b) { Dual<T, N> r = -b; r.v = a - b.v; return r; }
template<typename T, int N>
inline Dual<T, N> operator*(const Dual<T, N> &a, const T b) {
    Dual<T, N> r; r.v = a.v*b;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i]*b;
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator*(const T a, const Dual<T, N> &b) { return b*a; }
template<typename T, int N>
inline Dual<T, N> operator/(const Dual<T, N> &a, const T b) {
    Dual<T, N> r; r.v = a.v/b;
    for (int i = 0; i < N; i++) r.d[i] = a.d[i]/b;
    return r;
}
template<typename T, int N>
inline Dual<T, N> operator/(const T a, const Dual<T, N> &b) { return Dual<T, N>(a)/b; }

template<typename T, int N>
inline Dual<T, N> sqrt(const Dual<T, N> &a) {
    Dual<T, N> r; r.v = std::sqrt(a.v);
//  d sqrt(a) = da/(2 sqrt(a))   (sqrt_node::df, math.hpp:26 ff.)
    for (int i = 0; i < N; i++) r.d[i] = a.d[i]/(static_cast<T> (2)*r.v);
    return r;
}
//  fma_node (arithmetic.hpp:3736, compile :5079-5127: one real fma for real T).
template<typename T, int N>
inline Dual<T, N> fma(const Dual<T, N> &a, const Dual<T, N> &b, const Dual<T, N> &c) {
    Dual<T, N> r; r.v = std::fma(a.v, b.v, c.v);
    for (int i = 0; i < N; i++) r.d[i] = std::fma(a.d[i], b.v, std::fma(a.v, b.d[i], c.d[i]));
    return r;
}
template<typename T, int N>
inline Dual<T, N> fma(const T a, const Dual<T, N> &b, const T c) {
    Dual<T, N> r; r.v = std::fma(a, b.v, c);
    for (int i = 0; i < N; i++) r.d[i] = a*b.d[i];
    return r;
}
template<typename T, int N>
inline Dual<T, N> fma(const Dual<T, N> &a, const Dual<T, N> &b, const T c) {
    Dual<T, N> r; r.v = std::fma(a.v, b.v, c);
    for (int i = 0; i < N; i++) r.d[i] = std::fma(a.d[i], b.v, a.v*b.d[i]);
    return r;
}
inline double fma(const double a, const double b, const double c) { return std::fma(a, b, c); }
inline float fma(const float a, const float b, const float c) { return std::fma(a, b, c); }
inline double sqrt(const double a) { return std::sqrt(a); }
inline float sqrt(const float a) { return std::sqrt(a); }

template<typename T> inline T value_of(const T a) { return a; }
template<typename T, int N> inline T value_of(const Dual<T, N> &a) { return a.v; }

template<typename S> struct scalar_of { typedef S type; };
template<typename T, int N> struct scalar_of<Dual<T, N>> { typedef T type; };

// ---------------------------------------------------------------------------
// EFIT equilibrium data, exactly what equilibrium::make_efit reads
// (equilibrium.hpp:1628-1854): scalars cast to T (:1797-1805), coefficient
// tables cast to T (:1807-1842), num_cols = numz (:1849).
// Includes the constructor's member-init bug ne_c0(te_c0), ne_c1(te_c1)
// (equilibrium.hpp:1478) — reproduced in efit<T>::efit below.
// ---------------------------------------------------------------------------
template<typename T>
struct efit {
    T rmin, dr, zmin, dz, psimin, dpsi;
    T ne_scale, te_scale, pres_scale;
    size_t numr, numz, numpsi;
    std::vector<T> psi_c[4][4];   // psi_c[a][b] <-> file "psi_c<a><b>", [r index*numz + z index]
    std::vector<T> te_c[4], ne_c[4], pres_c[4], fpol_c[4];

    efit(const double *scalars9, const size_t numr_, const size_t numz_, const size_t numpsi_,
         const double *psi16, const double *te4, const double *ne4,
         const double *pres4, const double *fpol4) :
    rmin(static_cast<T> (scalars9[0])), dr(static_cast<T> (scalars9[1])),
    zmin(static_cast<T> (scalars9[2])), dz(static_cast<T> (scalars9[3])),
    psimin(static_cast<T> (scalars9[4])), dpsi(static_cast<T> (scalars9[5])),
    ne_scale(static_cast<T> (scalars9[6])), te_scale(static_cast<T> (scalars9[7])),
    pres_scale(static_cast<T> (scalars9[8])),
    numr(numr_), numz(numz_), numpsi(numpsi_) {
        for (int a = 0; a < 4; a++) {
            for (int b = 0; b < 4; b++) {
                const double *src = psi16 + (a*4 + b)*numr*numz;
                psi_c[a][b].assign(src, src + numr*numz);
            }
            te_c[a].assign(te4 + a*numpsi, te4 + (a + 1)*numpsi);
            ne_c[a].assign(ne4 + a*numpsi, ne4 + (a + 1)*numpsi);
            pres_c[a].assign(pres4 + a*numpsi, pres4 + (a + 1)*numpsi);
            fpol_c[a].assign(fpol4 + a*numpsi, fpol4 + (a + 1)*numpsi);
        }
//  equilibrium.hpp:1478 — ne_c0(te_c0), ne_c1(te_c1).  Part of the contract.
        ne_c[0] = te_c[0];
        ne_c[1] = te_c[1];
    }
};

// ---------------------------------------------------------------------------
// compile_index, piecewise.hpp:26-65:
//   (uint)min(max((x - offset)/scale, 0), length - 1)     truncation toward 0.
// The derivative of a gather is zero (piecewise.hpp:241-243, :947-949), so
// only the VALUE of the argument selects the cell.
// ---------------------------------------------------------------------------
template<typename T>
inline size_t gather_index(const T x, const T scale, const T offset, const size_t length) {
    const T q = (x - offset)/scale;
    const T c = std::min(std::max(q, static_cast<T> (0)), static_cast<T> (length - 1));
    return static_cast<size_t> (c);
}

// ---------------------------------------------------------------------------
// build_1D_spline, equilibrium.hpp:1121-1131.  c[k] are the gathered
// (index-shifted, piecewise.hpp:79-99) coefficients; scale/offset are folded
// into them and the cubic is evaluated in the RAW argument with three fmas.
// ---------------------------------------------------------------------------
template<typename S, typename T>
inline S build_1D_spline(const T c[4], const S &x, const T scale, const T offset) {
    const T three = static_cast<T> (3.0);
    const T two = static_cast<T> (2.0);
    const T s2 = scale*scale;
    const T s3 = scale*scale*scale;
    const T c3 = c[3]/s3;
    const T c2 = c[2]/s2 - three*offset*c[3]/s3;
    const T c1 = c[1]/scale - two*offset*c[2]/s2 + three*offset*offset*c[3]/s3;
    const T c0 = c[0] - offset*c[1]/scale + offset*offset*c[2]/s2 - offset*offset*offset*c[3]/s3;
    return fma(fma(fma(c3, x, c2), x, c1), x, c0);
}

// d/dx of the same expression, as symbolic df of the nested fma gives it:
//   a = fma(fma(c3,x,c2),x,c1);  b = fma(c3,x,c2)
//   d/dx fma(a,x,c0) = a + x*(b + x*c3)
template<typename S, typename T>
inline S build_1D_spline_dx(const T c[4], const S &x, const T scale, const T offset) {
    const T three = static_cast<T> (3.0);
    const T two = static_cast<T> (2.0);
    const T s2 = scale*scale;
    const T s3 = scale*scale*scale;
    const T c3 = c[3]/s3;
    const T c2 = c[2]/s2 - three*offset*c[3]/s3;
    const T c1 = c[1]/scale - two*offset*c[2]/s2 + three*offset*offset*c[3]/s3;
    const S b = fma(c3, x, c2);
    const S a = fma(b, x, c1);
    return a + x*(b + x*c3);
}

// ---------------------------------------------------------------------------
// efit::build_psi, equilibrium.hpp:1279-1313, plus its r and z derivatives
// (psi_cache->df(r), psi_cache->df(z) used at :1366, :1375).
// ---------------------------------------------------------------------------
template<typename S, typename T>
inline void build_psi(const efit<T> &eq, const S &r, const S &z,
                      S &psi, S &dpsi_dr, S &dpsi_dz) {
    const size_t i = gather_index(value_of(r), eq.dr, eq.rmin, eq.numr);
    const size_t j = gather_index(value_of(z), eq.dz, eq.zmin, eq.numz);
    const size_t cell = i*eq.numz + j;            // piecewise_2D: row*num_cols + col

    S C[4], dCdz[4];
    for (int a = 0; a < 4; a++) {
        const T c[4] = {eq.psi_c[a][0][cell], eq.psi_c[a][1][cell],
                        eq.psi_c[a][2][cell], eq.psi_c[a][3][cell]};
        C[a] = build_1D_spline(c, z, eq.dz, eq.zmin);          // :1307-1310
        dCdz[a] = build_1D_spline_dx(c, z, eq.dz, eq.zmin);
    }

    const S r_norm = (r - eq.rmin)/eq.dr;                       // :1305
    psi = ((C[3]*r_norm + C[2])*r_norm + C[1])*r_norm + C[0];   // :1312
    dpsi_dz = ((dCdz[3]*r_norm + dCdz[2])*r_norm + dCdz[1])*r_norm + dCdz[0];
//  d/dr of :1312 with d r_norm/dr = 1/dr.
    const S u = C[3]*r_norm + C[2];
    const S v = u*r_norm + C[1];
    dpsi_dr = (v + r_norm*(u + r_norm*C[3]))/eq.dr;
}

// 1D profile  scale*spline(c0..c3; psi)   (equilibrium.hpp:1338-1357, :1368-1373)
template<typename S, typename T>
inline S profile(const std::vector<T> c[4], const efit<T> &eq, const S &psi) {
    const size_t m = gather_index(value_of(psi), eq.dpsi, eq.psimin, eq.numpsi);
    const T cc[4] = {c[0][m], c[1][m], c[2][m], c[3][m]};
    return build_1D_spline(cc, psi, eq.dpsi, eq.psimin);
}

// ---------------------------------------------------------------------------
// efit::set_cache, equilibrium.hpp:1324-1384.
//   ne = ne_scale*spline(ne_c*)          :1343   (ne_c0/1 are te_c0/1, :1478)
//   te = te_scale*spline(te_c*)          :1350
//   ni = te                              :1361   (bug, part of the contract)
//   phi = atan(x, y) = atan2(y, x)       :1364   (cos = x/r, sin = y/r after reduce)
//   br = dpsi/dz / r ; bp = fpol/r ; bz = -dpsi/dr / r        :1366-1375
//   b = (br cos - bp sin, br sin + bp cos, bz)                :1380-1382
// ---------------------------------------------------------------------------
template<typename S>
struct field {
    S ne, te, ni, bx, by, bz, psi;
};

template<typename S, typename T>
inline field<S> set_cache(const efit<T> &eq, const S &x, const S &y, const S &z) {
    field<S> f;
    const S r = sqrt(x*x + y*y);                                // :1334
    S dpsi_dr, dpsi_dz;
    build_psi(eq, r, z, f.psi, dpsi_dr, dpsi_dz);               // :1336

    f.ne = eq.ne_scale*profile(eq.ne_c, eq, f.psi);
    f.te = eq.te_scale*profile(eq.te_c, eq, f.psi);
    f.ni = f.te;

    const S br = dpsi_dz/r;
    const S bp = profile(eq.fpol_c, eq, f.psi)/r;
    const S bz = -dpsi_dr/r;
    const S cosp = x/r;
    const S sinp = y/r;
    f.bx = br*cosp - bp*sinp;
    f.by = br*sinp + bp*cosp;
    f.bz = bz;
    return f;
}

// ---------------------------------------------------------------------------
// Physical constants, dispersion.hpp:490-503 (class physics) and the EFIT ion
// species equilibrium.hpp:1475 (deuterium mass 3.34449469E-27, charge 1).
// ---------------------------------------------------------------------------
template<typename T>
struct physics {
    const T epsilon0 = static_cast<T> (8.8541878138E-12);
    const T mu0 = static_cast<T> (M_PI*4.0E-7);
    const T q = static_cast<T> (1.602176634E-19);
    const T me = static_cast<T> (9.1093837015E-31);
    const T c = static_cast<T> (1.0)/std::sqrt(epsilon0*mu0);
    const T mi = static_cast<T> (3.34449469E-27);
    const T ion_charge = static_cast<T> (1);
};

// ---------------------------------------------------------------------------
// cold_plasma::D, dispersion.hpp:941-1008, statement by statement.
//   build_plasma_frequency   :326-332   n*q*q/(epsilon0*m*c*c)
//   build_cyclotron_frequency:348-353   q*b/(m*c)
// ---------------------------------------------------------------------------
template<typename S, typename T>
inline S cold_plasma_D(const efit<T> &eq, const S &w,
                       const S &kx, const S &ky, const S &kz,
                       const S &x, const S &y, const S &z) {
    const physics<T> ph;
    const T one = static_cast<T> (1.0);
    const field<S> f = set_cache(eq, x, y, z);

    const S wpe2 = f.ne*ph.q*ph.q/(ph.epsilon0*ph.me*ph.c*ph.c);            // :952-956
    const S b_len = sqrt(f.bx*f.bx + f.by*f.by + f.bz*f.bz);                // :958
    const S ec = (-ph.q)*b_len/(ph.me*ph.c);                                // :959-962

    const S w2 = w*w;                                                       // :964
    const S denome = one - ec*ec/w2;                                        // :965
    S e11 = one - (wpe2/w2)/denome;                                         // :966
    S e12 = ((ec/w)*(wpe2/w2))/denome;                                      // :967
    S e33 = wpe2;                                                           // :968

    {                                                                       // :970-986
        const T charge = ph.ion_charge*ph.q;
        const S wpi2 = f.ni*charge*charge/(ph.epsilon0*ph.mi*ph.c*ph.c);
        const S ic = charge*b_len/(ph.mi*ph.c);
        const S denomi = one - ic*ic/w2;
        e11 = e11 - (wpi2/w2)/denomi;
        e12 = e12 + ((ic/w)*(wpi2/w2))/denomi;
        e33 = e33 + wpi2;
    }

    e12 = static_cast<T> (-1.0)*e12;                                        // :988
    e33 = one - e33/w2;                                                     // :989

    const S nx = kx/w, ny = ky/w, nz = kz/w;                                // :992
    const S bhx = f.bx/b_len, bhy = f.by/b_len, bhz = f.bz/b_len;           // :993

    const S npara = bhx*nx + bhy*ny + bhz*nz;                               // :995
    const S npara2 = npara*npara;
    const S cx = bhy*nz - bhz*ny;                                           // :997 (vector.hpp cross)
    const S cy = bhz*nx - bhx*nz;
    const S cz = bhx*ny - bhy*nx;
    const S nperp = sqrt(cx*cx + cy*cy + cz*cz);
    const S nperp2 = nperp*nperp;

    const S m11 = e11 - npara2;                                             // :1001-1005
    const S m12 = e12;
    const S m13 = npara*nperp;
    const S m22 = e11 - npara2 - nperp2;
    const S m33 = e33 - nperp2;

    return (m11*m22 - m12*m12)*m33 - m22*(m13*m13);                         // :1007
}

// ---------------------------------------------------------------------------
// dispersion_interface ctor, dispersion.hpp:1369-1434.  EFIT is Cartesian
// (esup = identity, equilibrium.hpp:379-420) so k_vec = (kx,ky,kz), dk/dx = 0:
//   dxdt  = -dD/dkx / dD/dw                                   :1426-1428
//   dkxdt =  dD/dx  / dD/dw                                   :1429-1431
// Tangent slots: 0 w, 1 kx, 2 ky, 3 kz, 4 x, 5 y, 6 z.
// ---------------------------------------------------------------------------
template<typename T>
struct ray_rhs {
    T D;
    T dD[7];
    T dxdt, dydt, dzdt, dkxdt, dkydt, dkzdt;
};

template<typename T>
inline ray_rhs<T> dispersion_rhs(const efit<T> &eq, const T w,
                                 const T kx, const T ky, const T kz,
                                 const T x, const T y, const T z) {
    typedef Dual<T, 7> S;
    const S Dd = cold_plasma_D(eq, S::seed(w, 0),
                               S::seed(kx, 1), S::seed(ky, 2), S::seed(kz, 3),
                               S::seed(x, 4), S::seed(y, 5), S::seed(z, 6));
    ray_rhs<T> r;
    r.D = Dd.v;
    for (int i = 0; i < 7; i++) r.dD[i] = Dd.d[i];
    r.dxdt = -Dd.d[1]/Dd.d[0];
    r.dydt = -Dd.d[2]/Dd.d[0];
    r.dzdt = -Dd.d[3]/Dd.d[0];
    r.dkxdt = Dd.d[4]/Dd.d[0];
    r.dkydt = Dd.d[5]/Dd.d[0];
    r.dkzdt = Dd.d[6]/Dd.d[0];
    return r;
}

// SoA ray state as the reference's kernels see it: inputs
// {t, w, x, y, z, kx, ky, kz} (solver.hpp:304-313).
template<typename T>
struct rays {
    T *t, *w, *x, *y, *z, *kx, *ky, *kz;
};

// ---------------------------------------------------------------------------
// `loss_kernel`: solver::newton (newton.hpp:34-51) for one unknown:
//   setter  var <- var - step*D/(dD/dvar)      output  D*D   (both from the
//   INPUT state).   var: 0 w, 1 kx, 2 ky, 3 kz  (dispersion_test.cpp:26-64).
// ---------------------------------------------------------------------------
template<typename T>
inline void loss_kernel(const efit<T> &eq, const size_t n, rays<T> s, T *residual,
                        const int var, const T step) {
    for (size_t i = 0; i < n; i++) {
        const ray_rhs<T> r = dispersion_rhs(eq, s.w[i], s.kx[i], s.ky[i], s.kz[i],
                                            s.x[i], s.y[i], s.z[i]);
        T *target = var == 0 ? s.w : (var == 1 ? s.kx : (var == 2 ? s.ky : s.kz));
        target[i] = target[i] - step*r.D/r.dD[var];
        residual[i] = r.D*r.D;
    }
}

// ---------------------------------------------------------------------------
// converge_item::run, workflow.hpp:179-205, with the max reduction of
// cpu_context::create_max_call (cpu_context.hpp:306-322, std::max_element).
// Returns the number of loop iterations; *last receives the final max.
// ---------------------------------------------------------------------------
template<typename T>
inline size_t newton_solve(const efit<T> &eq, const size_t n, rays<T> s, T *residual,
                           const int var, const T step, const T tolerance,
                           const size_t max_iterations, T *last) {
    auto max_kernel = [&] () -> T {
        loss_kernel(eq, n, s, residual, var, step);
        return *std::max_element(residual, residual + n);
    };
    size_t iterations = 0;
    T max_residual = max_kernel();
    T last_max = std::numeric_limits<T>::max();
    T off_last_max = std::numeric_limits<T>::max();
    while (std::abs(max_residual) > std::abs(tolerance)                &&
           std::abs(last_max - max_residual) > std::abs(tolerance)     &&
           std::abs(off_last_max - max_residual) > std::abs(tolerance) &&
           iterations++ < max_iterations) {
        last_max = max_residual;
        if (!(iterations%2)) {
            off_last_max = max_residual;
        }
        max_residual = max_kernel();
    }
    *last = max_residual;
    return iterations;
}

// ---------------------------------------------------------------------------
// `solver_kernel`: solver::rk4 ctor, solver.hpp:777-870, with the item of
// solver_interface::compile (:303-349): output residual = D*D at the INPUT
// state (dispersion.hpp:1474), setters {kx,ky,kz,x,y,z,t}_next.
// ---------------------------------------------------------------------------
template<typename T>
inline void rk4_step(const efit<T> &eq, const size_t begin, const size_t end,
                     rays<T> s, T *residual, const T dt) {
    const T two = static_cast<T> (2.0);
    const T six = static_cast<T> (6.0);
    for (size_t i = begin; i < end; i++) {
        const T w = s.w[i], kx = s.kx[i], ky = s.ky[i], kz = s.kz[i];
        const T x = s.x[i], y = s.y[i], z = s.z[i], t = s.t[i];

        const ray_rhs<T> r1 = dispersion_rhs(eq, w, kx, ky, kz, x, y, z);
        const T kx1 = dt*r1.dkxdt, ky1 = dt*r1.dkydt, kz1 = dt*r1.dkzdt;    // :811-816
        const T x1 = dt*r1.dxdt, y1 = dt*r1.dydt, z1 = dt*r1.dzdt;

        const ray_rhs<T> r2 = dispersion_rhs(eq, w, kx + kx1/two, ky + ky1/two, kz + kz1/two,
                                             x + x1/two, y + y1/two, z + z1/two);   // :820-828
        const T kx2 = dt*r2.dkxdt, ky2 = dt*r2.dkydt, kz2 = dt*r2.dkzdt;
        const T x2 = dt*r2.dxdt, y2 = dt*r2.dydt, z2 = dt*r2.dzdt;

        const ray_rhs<T> r3 = dispersion_rhs(eq, w, kx + kx2/two, ky + ky2/two, kz + kz2/two,
                                             x + x2/two, y + y2/two, z + z2/two);   // :837-845
        const T kx3 = dt*r3.dkxdt, ky3 = dt*r3.dkydt, kz3 = dt*r3.dkzdt;
        const T x3 = dt*r3.dxdt, y3 = dt*r3.dydt, z3 = dt*r3.dzdt;

        const ray_rhs<T> r4 = dispersion_rhs(eq, w, kx + kx3, ky + ky3, kz + kz3,
                                             x + x3, y + y3, z + z3);               // :856-864
        const T kx4 = dt*r4.dkxdt, ky4 = dt*r4.dkydt, kz4 = dt*r4.dkzdt;
        const T x4 = dt*r4.dxdt, y4 = dt*r4.dydt, z4 = dt*r4.dzdt;

        residual[i] = r1.D*r1.D;
        s.kx[i] = kx + (kx1 + two*(kx2 + kx3) + kx4)/six;                   // :864-869
        s.ky[i] = ky + (ky1 + two*(ky2 + ky3) + ky4)/six;
        s.kz[i] = kz + (kz1 + two*(kz2 + kz3) + kz4)/six;
        s.x[i]  = x  + (x1  + two*(x2  + x3 ) + x4 )/six;
        s.y[i]  = y  + (y1  + two*(y2  + y3 ) + y4 )/six;
        s.z[i]  = z  + (z1  + two*(z2  + z3 ) + z4 )/six;
        s.t[i]  = t + dt;                                                   // :854
    }
}

// ---------------------------------------------------------------------------
// efit_test's `test_kernel`, graph_tests/efit_test.cpp:145-162: outputs
// bx, by, bz, ne, te and div B = dbx/dx + dby/dy + dbz/dz.
// ---------------------------------------------------------------------------
template<typename T>
inline void efit_test_kernel(const efit<T> &eq, const size_t n,
                             const T *x, const T *y, const T *z,
                             T *bx, T *by, T *bz, T *ne, T *te, T *div) {
    typedef Dual<T, 3> S;
    for (size_t i = 0; i < n; i++) {
        const field<S> f = set_cache(eq, S::seed(x[i], 0), S::seed(y[i], 1), S::seed(z[i], 2));
        bx[i] = f.bx.v; by[i] = f.by.v; bz[i] = f.bz.v;
        ne[i] = f.ne.v; te[i] = f.te.v;
        div[i] = f.bx.d[0] + f.by.d[1] + f.bz.d[2];
    }
}

// ---------------------------------------------------------------------------
// efit::get_characteristic_field, equilibrium.hpp:1585-1615: two-unknown
// Newton (x and z, both updated from the same input state, step 0.1) on
// f = (psi - psimin)/dpsi to reach the magnetic axis, then |B| there
// (`bmod_at_axis`).  Single element, host loop of workflow.hpp:179-205.
// ---------------------------------------------------------------------------
template<typename T>
inline T characteristic_field(const efit<T> &eq, size_t *iterations_out) {
    typedef Dual<T, 2> S;
    T x = static_cast<T> (1.7), y = static_cast<T> (0.0), z = static_cast<T> (0.0);
    const T step = static_cast<T> (0.1);
    const T tolerance = static_cast<T> (1.0E-30);
    const size_t max_iterations = 1000;

    auto max_kernel = [&] () -> T {
        const S xs = S::seed(x, 0), ys(y), zs = S::seed(z, 1);
        const field<S> f = set_cache(eq, xs, ys, zs);
        const S func = (f.psi - eq.psimin)/eq.dpsi;
        const T x_new = x - step*func.v/func.d[0];
        const T z_new = z - step*func.v/func.d[1];
        x = x_new;
        z = z_new;
        return func.v*func.v;
    };
    size_t iterations = 0;
    T max_residual = max_kernel();
    T last_max = std::numeric_limits<T>::max();
    T off_last_max = std::numeric_limits<T>::max();
    while (std::abs(max_residual) > std::abs(tolerance)                &&
           std::abs(last_max - max_residual) > std::abs(tolerance)     &&
           std::abs(off_last_max - max_residual) > std::abs(tolerance) &&
           iterations++ < max_iterations) {
        last_max = max_residual;
        if (!(iterations%2)) {
            off_last_max = max_residual;
        }
        max_residual = max_kernel();
    }
    if (iterations_out) *iterations_out = iterations;

    const field<T> f = set_cache(eq, x, y, z);
    return std::sqrt(f.bx*f.bx + f.by*f.by + f.bz*f.bz);
}

// ---------------------------------------------------------------------------
// xkorc, graph_korc/xkorc.cpp:29-121.
//   initialize_gamma (:66-85): gamma = 1/sqrt(1 - u.u); u <- gamma*u
//   step (:87-121), dt = 0.5, b = B(pos)/b0, larmor_radius = c*me/(q*b0)
// ---------------------------------------------------------------------------
template<typename T>
struct korc_constants {
    T b0, larmor_radius, dt;
    korc_constants(const T b0_) : b0(b0_), dt(static_cast<T> (0.5)) {
        const T q = static_cast<T> (1.602176634E-19);                       // :32-34
        const T me = static_cast<T> (9.1093837139E-31);
        const T c = static_cast<T> (299792458.0);
        const T gyro_period = me/(q*b0);                                    // :36
        larmor_radius = c*gyro_period;                                      // :38
    }
};

template<typename T>
inline void korc_initialize_gamma(const size_t n, T *ux, T *uy, T *uz, T *gamma) {
    const T one = static_cast<T> (1.0);
    for (size_t i = 0; i < n; i++) {
        const T g = one/std::sqrt(one - (ux[i]*ux[i] + uy[i]*uy[i] + uz[i]*uz[i]));
        ux[i] = g*ux[i];
        uy[i] = g*uy[i];
        uz[i] = g*uz[i];
        gamma[i] = g;
    }
}

template<typename T>
inline void korc_step(const efit<T> &eq, const korc_constants<T> &k,
                      const size_t begin, const size_t end,
                      T *x, T *y, T *z, T *ux, T *uy, T *uz, T *gamma) {
    const T one = static_cast<T> (1.0), two = static_cast<T> (2.0);
    const T half = static_cast<T> (0.5), four = static_cast<T> (4.0);
    const T dt = k.dt;
    for (size_t i = begin; i < end; i++) {
        const field<T> f = set_cache(eq, x[i], y[i], z[i]);
        const T bx = f.bx/k.b0, by = f.by/k.b0, bz = f.bz/k.b0;            // :70-72
        const T vx = ux[i], vy = uy[i], vz = uz[i], g = gamma[i];

//  u' = u - dt*(u x b)/(2 gamma)                                           :87
        const T cx = vy*bz - vz*by, cy = vz*bx - vx*bz, cz = vx*by - vy*bx;
        const T px = vx - dt*cx/(two*g), py = vy - dt*cy/(two*g), pz = vz - dt*cz/(two*g);

        const T tx = -half*dt*bx, ty = -half*dt*by, tz = -half*dt*bz;     // :89
        const T tau_sq = tx*tx + ty*ty + tz*tz;                            // :90
        const T speed_sq = px*px + py*py + pz*pz;                          // :91
        const T sigma = one + speed_sq - tau_sq;                           // :92
        const T ustar = px*tx + py*ty + pz*tz;                             // :93

        const T gamma_next = std::sqrt(half*(sigma + std::sqrt(sigma*sigma + four*(tau_sq + ustar*ustar))));  // :95
        const T sx = tx/gamma_next, sy = ty/gamma_next, sz = tz/gamma_next; // :96  t = tau/gamma_next

        const T s = one + (sx*sx + sy*sy + sz*sz);                         // :98
        const T updt = px*sx + py*sy + pz*sz;                              // :99

        const T qx = py*sz - pz*sy, qy = pz*sx - px*sz, qz = px*sy - py*sx; // u' x t
        const T nux = (px + updt*sx + qx)/s;                               // :101
        const T nuy = (py + updt*sy + qy)/s;
        const T nuz = (pz + updt*sz + qz)/s;

        x[i] = x[i] + k.larmor_radius*dt*nux/gamma_next;                   // :103
        y[i] = y[i] + k.larmor_radius*dt*nuy/gamma_next;
        z[i] = z[i] + k.larmor_radius*dt*nuz/gamma_next;
        ux[i] = nux; uy[i] = nuy; uz[i] = nuz;
        gamma[i] = gamma_next;
    }
}

}  // namespace gfo

#endif /* GF_ORACLE_HPP */

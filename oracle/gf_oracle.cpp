// ---------------------------------------------------------------------------
// gf_oracle.cpp — C entry points of the CPU ORACLE (test infrastructure only).
//
// Loaded with ctypes by tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py.  Never linked or loaded by graph_framework_amd's product
// path.  See gf_oracle.hpp for the reference citations.
// ---------------------------------------------------------------------------
#include <limits>
#include <thread>
#include <chrono>
#include <cstring>
#include "gf_oracle.hpp"

namespace {

template<typename T>
gfo::rays<T> make_rays(T *t, T *w, T *x, T *y, T *z, T *kx, T *ky, T *kz) {
    gfo::rays<T> s;
    s.t = t; s.w = w; s.x = x; s.y = y; s.z = z; s.kx = kx; s.ky = ky; s.kz = kz;
    return s;
}

//  Shard split of graph_benchmark/xrays_bench.cpp:38-51:
//  batch = N/T (+1 for the first N%T threads), contiguous.
inline void shard(const size_t n, const size_t threads, const size_t index,
                  size_t &begin, size_t &end) {
    const size_t batch = n/threads;
    const size_t extra = n%threads;
    begin = index*batch + std::min(index, extra);
    end = begin + batch + (extra > index ? 1 : 0);
}

template<typename T>
double rk4_steps(const gfo::efit<T> &eq, const size_t n, gfo::rays<T> s, T *residual,
                 const T dt, const size_t num_steps, size_t threads) {
    threads = std::max<size_t> (1, std::min(threads, n));
    const auto start = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (size_t i = 0; i < threads; i++) {
        pool.emplace_back([&, i] {
            size_t begin, end;
            shard(n, threads, i, begin, end);
            for (size_t j = 0; j < num_steps; j++) {
                gfo::rk4_step(eq, begin, end, s, residual, dt);
            }
        });
    }
    for (auto &t : pool) t.join();
    const auto stop = std::chrono::steady_clock::now();
    return std::chrono::duration<double> (stop - start).count();
}

template<typename T>
double korc_steps(const gfo::efit<T> &eq, const T b0, const size_t n,
                  T *x, T *y, T *z, T *ux, T *uy, T *uz, T *gamma,
                  const size_t num_steps, size_t threads) {
    const gfo::korc_constants<T> k(b0);
    threads = std::max<size_t> (1, std::min(threads, n));
    const auto start = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (size_t i = 0; i < threads; i++) {
        pool.emplace_back([&, i] {
            size_t begin, end;
            shard(n, threads, i, begin, end);
            for (size_t j = 0; j < num_steps; j++) {
                gfo::korc_step(eq, k, begin, end, x, y, z, ux, uy, uz, gamma);
            }
        });
    }
    for (auto &t : pool) t.join();
    const auto stop = std::chrono::steady_clock::now();
    return std::chrono::duration<double> (stop - start).count();
}

}  // namespace

#define GFO_DEFINE(SUFFIX, T)                                                              \
extern "C" void *gfo_efit_create_##SUFFIX(const double *scalars9, const size_t numr,       \
                                          const size_t numz, const size_t numpsi,          \
                                          const double *psi16, const double *te4,          \
                                          const double *ne4, const double *pres4,          \
                                          const double *fpol4) {                           \
    return new gfo::efit<T> (scalars9, numr, numz, numpsi, psi16, te4, ne4, pres4, fpol4); \
}                                                                                          \
extern "C" void gfo_efit_destroy_##SUFFIX(void *eq) {                                      \
    delete static_cast<gfo::efit<T> *> (eq);                                               \
}                                                                                          \
extern "C" void gfo_efit_test_kernel_##SUFFIX(const void *eq, const size_t n,              \
                                              const T *x, const T *y, const T *z,          \
                                              T *bx, T *by, T *bz, T *ne, T *te, T *div) { \
    gfo::efit_test_kernel(*static_cast<const gfo::efit<T> *> (eq), n, x, y, z,             \
                          bx, by, bz, ne, te, div);                                        \
}                                                                                          \
/*  D and its 7 partials (slot order w,kx,ky,kz,x,y,z), dD is [7][n].  */                   \
extern "C" void gfo_cold_plasma_D_##SUFFIX(const void *eq, const size_t n, const T *w,     \
                                           const T *kx, const T *ky, const T *kz,          \
                                           const T *x, const T *y, const T *z,             \
                                           T *D, T *dD) {                                  \
    for (size_t i = 0; i < n; i++) {                                                       \
        const gfo::ray_rhs<T> r = gfo::dispersion_rhs(*static_cast<const gfo::efit<T> *> (eq), \
                                                      w[i], kx[i], ky[i], kz[i], x[i], y[i], z[i]); \
        D[i] = r.D;                                                                        \
        for (int j = 0; j < 7; j++) dD[j*n + i] = r.dD[j];                                 \
    }                                                                                      \
}                                                                                          \
extern "C" void gfo_loss_kernel_##SUFFIX(const void *eq, const size_t n,                   \
                                         T *t, T *w, T *x, T *y, T *z, T *kx, T *ky, T *kz,\
                                         T *residual, const int var, const T step) {       \
    gfo::loss_kernel(*static_cast<const gfo::efit<T> *> (eq), n,                           \
                     make_rays(t, w, x, y, z, kx, ky, kz), residual, var, step);           \
}                                                                                          \
extern "C" size_t gfo_newton_solve_##SUFFIX(const void *eq, const size_t n,                \
                                            T *t, T *w, T *x, T *y, T *z,                  \
                                            T *kx, T *ky, T *kz, T *residual,              \
                                            const int var, const T step, const T tolerance,\
                                            const size_t max_iterations, T *last_max) {    \
    return gfo::newton_solve(*static_cast<const gfo::efit<T> *> (eq), n,                   \
                             make_rays(t, w, x, y, z, kx, ky, kz), residual, var, step,    \
                             tolerance, max_iterations, last_max);                         \
}                                                                                          \
/*  num_steps RK4 steps on `threads` host threads; returns wall seconds.  */               \
extern "C" double gfo_rk4_steps_##SUFFIX(const void *eq, const size_t n,                   \
                                         T *t, T *w, T *x, T *y, T *z,                     \
                                         T *kx, T *ky, T *kz, T *residual, const T dt,     \
                                         const size_t num_steps, const size_t threads) {   \
    return rk4_steps(*static_cast<const gfo::efit<T> *> (eq), n,                           \
                     make_rays(t, w, x, y, z, kx, ky, kz), residual, dt, num_steps,        \
                     threads);                                                             \
}                                                                                          \
extern "C" T gfo_characteristic_field_##SUFFIX(const void *eq, size_t *iterations) {       \
    return gfo::characteristic_field(*static_cast<const gfo::efit<T> *> (eq), iterations); \
}                                                                                          \
extern "C" void gfo_korc_constants_##SUFFIX(const T b0, T *larmor_radius, T *dt) {         \
    const gfo::korc_constants<T> k(b0);                                                    \
    *larmor_radius = k.larmor_radius;                                                      \
    *dt = k.dt;                                                                            \
}                                                                                          \
extern "C" void gfo_korc_initialize_gamma_##SUFFIX(const size_t n, T *ux, T *uy, T *uz,    \
                                                   T *gamma) {                             \
    gfo::korc_initialize_gamma(n, ux, uy, uz, gamma);                                      \
}                                                                                          \
extern "C" double gfo_korc_steps_##SUFFIX(const void *eq, const T b0, const size_t n,      \
                                          T *x, T *y, T *z, T *ux, T *uy, T *uz, T *gamma, \
                                          const size_t num_steps, const size_t threads) {  \
    return korc_steps(*static_cast<const gfo::efit<T> *> (eq), b0, n, x, y, z,             \
                      ux, uy, uz, gamma, num_steps, threads);                              \
}

GFO_DEFINE(f64, double)
GFO_DEFINE(f32, float)

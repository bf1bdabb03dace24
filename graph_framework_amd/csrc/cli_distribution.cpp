//------------------------------------------------------------------------------
///  @file cli_distribution.cpp
///  @brief The initial conditions of the xrays command line, sample for sample.
///
///  graph_driver/xrays.cpp:413-453 seeds one std::mt19937_64 per device thread with the thread
///  index (make_engine, :397-399, with --seed) and draws, in this order, omega, kx, ky, kz, z and
///  (x, y) for ALL rays of the shard, each through its own std::normal_distribution
///  (set_variable :57-74, set_xy_variables :81-131).  std::normal_distribution's algorithm is not
///  fixed by the standard; the reference's numbers are libstdc++'s: Marsaglia's polar method on
///  std::generate_canonical<double, 53> (one 64-bit word per uniform), the second value of each
///  pair saved for the next call.  Both are restated here by hand (no <random>), so that the
///  samples do not depend on the standard library this file is built with;
///  tests/golden/cli_distribution_golden.npz holds samples of the real libstdc++ objects.
//------------------------------------------------------------------------------
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace {

//  std::mt19937_64 (Matsumoto & Nishimura's MT19937-64).
struct mt19937_64 {
    uint64_t state[312];
    size_t position;

    explicit mt19937_64(const uint64_t seed) {
        state[0] = seed;
        for (size_t i = 1; i < 312; i++) {
            state[i] = 6364136223846793005ull*(state[i - 1]^(state[i - 1] >> 62)) + i;
        }
        position = 312;
    }

    uint64_t operator()() {
        if (position >= 312) {
            for (size_t i = 0; i < 312; i++) {
                const uint64_t x = (state[i] & 0xFFFFFFFF80000000ull) | (state[(i + 1)%312] & 0x7FFFFFFFull);
                state[i] = state[(i + 156)%312]^(x >> 1)^((x & 1ull) ? 0xB5026F5AA96619E9ull : 0ull);
            }
            position = 0;
        }
        uint64_t y = state[position++];
        y ^= (y >> 29) & 0x5555555555555555ull;
        y ^= (y << 17) & 0x71D67FFFEDA60000ull;
        y ^= (y << 37) & 0xFFF7EEE000000000ull;
        y ^= y >> 43;
        return y;
    }
};

//  std::generate_canonical<double, 53> over a 64-bit engine: one word, rounded to double, / 2^64.
double canonical(mt19937_64 &engine) {
    const double value = static_cast<double> (engine())/18446744073709551616.0;
    return value >= 1.0 ? std::nextafter(1.0, 0.0) : value;
}

//  libstdc++'s std::normal_distribution<double>::operator().
struct normal_distribution {
    double mean, sigma, saved;
    bool saved_available;

    normal_distribution(const double m, const double s) : mean(m), sigma(s), saved(0.0), saved_available(false) {}

    double operator()(mt19937_64 &engine) {
        double value;
        if (saved_available) {
            saved_available = false;
            value = saved;
        } else {
            double x, y, r2;
            do {
                x = 2.0*canonical(engine) - 1.0;
                y = 2.0*canonical(engine) - 1.0;
                r2 = x*x + y*y;
            } while (r2 > 1.0 || r2 == 0.0);
            const double mult = std::sqrt(-2.0*std::log(r2)/r2);
            saved = x*mult;
            saved_available = true;
            value = y*mult;
        }
        return value*sigma + mean;
    }
};

}  // namespace

//  columns[0..7] = t, w, x, y, z, kx, ky, kz (the input order of solver_interface), n doubles each.
//  means/sigmas in the draw order omega, kx, ky, kz, z, radius, phi; sigma <= 0 = "not normal": the mean.
//  Cylindrical (x, y) as `--use_cyl_xy` with a normal angle and a fixed radius
//  (xrays.cpp:109-118: one draw per ray), or both normal (:93-104: two draws per ray, BOTH from the
//  angle's distribution — the reference's own code, kept).
extern "C" void gfhip_cli_distribution(const uint64_t seed, const size_t n, const double *means, const double *sigmas,
                                       double *const *columns) {
    mt19937_64 engine(seed);
    for (size_t i = 0; i < n; i++) columns[0][i] = 0.0;
    const int order[5] = {1, 5, 6, 7, 4};                   // omega, kx, ky, kz, z -> column
    for (int v = 0; v < 5; v++) {
        if (sigmas[v] > 0.0) {
            normal_distribution distribution(means[v], sigmas[v]);
            for (size_t i = 0; i < n; i++) columns[order[v]][i] = distribution(engine);
        } else {
            for (size_t i = 0; i < n; i++) columns[order[v]][i] = means[v];
        }
    }
    const double radius_mean = means[5], phi_mean = means[6];
    if (sigmas[5] > 0.0 && sigmas[6] > 0.0) {
        normal_distribution phi_distribution(phi_mean, sigmas[6]);
        for (size_t i = 0; i < n; i++) {
            const double r = phi_distribution(engine);
            const double phi = phi_distribution(engine);
            columns[2][i] = r*std::cos(phi);
            columns[3][i] = r*std::sin(phi);
        }
    } else if (sigmas[6] > 0.0) {
        normal_distribution phi_distribution(phi_mean, sigmas[6]);
        for (size_t i = 0; i < n; i++) {
            const double phi = phi_distribution(engine);
            columns[2][i] = radius_mean*std::cos(phi);
            columns[3][i] = radius_mean*std::sin(phi);
        }
    } else {
        for (size_t i = 0; i < n; i++) {
            columns[2][i] = radius_mean*std::cos(phi_mean);
            columns[3][i] = radius_mean*std::sin(phi_mean);
        }
    }
}

// Samples of the REAL libstdc++ objects the reference draws its CLI initial conditions with
// (graph_driver/xrays.cpp:397-453: std::mt19937_64 seeded with the shard index, one
// std::normal_distribution per variable), in the example's configuration
// (graph_driver/CMakeLists.txt:5-31).  Prints, per seed, the first `count` samples of
// omega, ky, kz, z and phi as hexadecimal doubles; tests/golden/make_cli_fixture.py stores them in
// cli_distribution_golden.npz.      g++ -O1 -o make_cli_fixture make_cli_fixture.cpp
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000;
    const size_t count = argc > 2 ? strtoull(argv[2], nullptr, 10) : 16;
    for (unsigned long seed : {0ul, 1ul, 7ul}) {
        std::mt19937_64 engine(seed);
        const double means[5] = {700.0, -100.0, 0.0, 0.0, 0.0}, sigmas[5] = {10.0, 10.0, 10.0, 0.05, 0.05};
        for (int v = 0; v < 5; v++) {                      // omega, ky, kz, z, phi (kx is a constant: no draw)
            std::normal_distribution<double> distribution(means[v], sigmas[v]);
            for (size_t i = 0; i < n; i++) {
                const double value = distribution(engine);
                if (i < count || i + 1 == n) printf("%lu %d %zu %a\n", seed, v, i, value);
            }
        }
    }
    return 0;
}

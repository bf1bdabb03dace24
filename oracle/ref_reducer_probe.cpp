// ---------------------------------------------------------------------------
// ref_reducer_probe.cpp — TEST INFRASTRUCTURE (oracle/Makefile target _ref/ref_reducer_probe; needs
// /root/reference; runs on the CPU: tests/test_oracle.py::test_reference_reducer_does_not_get_through_vmec).
//
// Why the VMEC ray equations are not built: differentiating a dispersion function
// with respect to the flux coordinate s sends the reference's own algebraic reducer into rewrite
// recursions that do not return.  This program shows it WITHOUT the restated equilibrium class of
// ref_builders.hpp: main() below calls the reference's node factories only (graph::variable,
// graph::constant, graph::piecewise_1D, graph::fma, graph::sin/cos, graph::vector, graph::matrix, the
// operators, vector_quantity::dot/cross/unit, df), statement for statement what
// equilibrium::vmec::set_cache (equilibrium.hpp:2073-2140, get_esubs/u/v :1920-2013, get_chi/get_phi
// :2036-2059, build_1D_spline :1121-1131) and dispersion::cold_plasma::D (dispersion.hpp:995-1001)
// execute, on the spline tables of the reference-held graph_tests/vmec.nc (written to a flat file by
// tests/golden/make_vmec_golden.py::write_vmec).
//
//   ref_reducer_probe <vmec.bin> <modes> parts    d/ds of B_x, |B|, b_x, k_x, n.b, n.n: each returns at once
//   ref_reducer_probe <vmec.bin> <modes> cross    d/ds of (b x n)_x: the smallest expression found that does
//                                                 not return (add_node::reduce, arithmetic.hpp:296 <->
//                                                 subtract_node::reduce, :1242); run it under `timeout`
//   ref_reducer_probe synthetic                   no file: TWO modes on made-up tables, (m, n) = (0, 0), (2, 3);
//                                                 then dR/du is ONE product, and cos(v)*dR/du alone — the first
//                                                 component of e_u — overflows the stack inside
//                                                 multiply_node::reduce (arithmetic.hpp:2006 "(a*v)*b -> (a*b)*v"
//                                                 <-> :2076 "(a*cos)*b -> (a*b)*cos"): a second rewrite cycle
//                                                 of the same kind, within milliseconds
// ---------------------------------------------------------------------------
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "node.hpp"
#include "arithmetic.hpp"
#include "math.hpp"
#include "trigonometry.hpp"
#include "piecewise.hpp"
#include "vector.hpp"

typedef double T;
typedef graph::shared_leaf<T> leaf;
typedef graph::shared_vector<T> vec3;

static double seconds() {
    return std::chrono::duration<double> (std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void timed(const char *label, leaf expression, leaf s) {
    const double t0 = seconds();
    leaf derivative = expression->df(s);
    std::printf("d/ds %-8s returned in %.3f s\n", label, seconds() - t0);
    std::fflush(stdout);
    (void)derivative;
}

//  The flat file of make_vmec_golden.py::write_vmec: sminh sminf ds dphi signj | numsf numsh nummn |
//  chi_c0..3 [numsf] | rmnc_c0..3, zmns_c0..3 [nummn][numsf], lmns_c0..3 [nummn][numsh] | xm | xn.
struct tables {
    double sminh, sminf, ds, dphi, signj;
    uint64_t numsf = 9, numsh = 9, nummn = 2;
    std::vector<double> chi[4], xm, xn;
    std::vector<std::vector<double>> rmnc[4], zmns[4], lmns[4];

    void read(const char *path) {
        FILE *f = std::fopen(path, "rb");
        if (!f) { std::perror(path); std::exit(1); }
        auto rd = [f] (void *p, size_t bytes) { if (std::fread(p, 1, bytes, f) != bytes) { std::fprintf(stderr, "short read\n"); std::exit(1); } };
        double scalars[5];
        rd(scalars, sizeof(scalars));
        sminh = scalars[0]; sminf = scalars[1]; ds = scalars[2]; dphi = scalars[3]; signj = scalars[4];
        rd(&numsf, 8); rd(&numsh, 8); rd(&nummn, 8);
        for (auto &c : chi) { c.resize(numsf); rd(c.data(), 8*numsf); }
        for (auto *q : {rmnc, zmns, lmns}) {
            const size_t length = q == lmns ? numsh : numsf;
            for (int k = 0; k < 4; k++) {
                q[k].assign(nummn, std::vector<double> (length));
                for (auto &row : q[k]) rd(row.data(), 8*length);
            }
        }
        xm.resize(nummn); rd(xm.data(), 8*nummn);
        xn.resize(nummn); rd(xn.data(), 8*nummn);
        std::fclose(f);
    }

//  Made-up tables for `synthetic`: nothing depends on their values.
    void synthetic() {
        sminh = 0.0625; sminf = 0.0; ds = 0.125; dphi = 0.8; signj = -1.0;
        auto column = [] (const double base, const double slope) {
            std::vector<double> values(9);
            for (size_t i = 0; i < values.size(); i++) values[i] = base + slope*static_cast<double> (i) + 0.01*static_cast<double> ((i*7 + 3)%5);
            return values;
        };
        for (int k = 0; k < 4; k++) {
            chi[k] = column(0.2/(k + 1), 0.03);
            rmnc[k] = {column(1.5/(k + 1), 0.01), column(0.4/(k + 1), 0.05)};
            zmns[k] = {column(0.0, 0.0), column(0.5/(k + 1), 0.06)};
            lmns[k] = {column(0.0, 0.0), column(0.1/(k + 1), 0.01)};
        }
        xm = {0.0, 2.0};
        xn = {0.0, 3.0};
    }
};

static backend::buffer<T> to_buffer(const std::vector<double> &v) {
    return backend::buffer<T> (std::vector<T> (v.begin(), v.end()));
}

//  equilibrium.hpp:1121-1131
static leaf build_1D_spline(std::vector<leaf> c, leaf x, const T scale, const T offset) {
    auto c3 = c[3]/(scale*scale*scale);
    auto c2 = c[2]/(scale*scale) - static_cast<T> (3.0)*offset*c[3]/(scale*scale*scale);
    auto c1 = c[1]/scale - static_cast<T> (2.0)*offset*c[2]/(scale*scale) + static_cast<T> (3.0)*offset*offset*c[3]/(scale*scale*scale);
    auto c0 = c[0] - offset*c[1]/scale + offset*offset*c[2]/(scale*scale) - offset*offset*offset*c[3]/(scale*scale*scale);
    return graph::fma(graph::fma(graph::fma(c3, x, c2), x, c1), x, c0);
}

int main(int argc, char **argv) {
    const bool synthetic = argc == 2 && !std::strcmp(argv[1], "synthetic");
    if (!synthetic && (argc != 4 || (std::strcmp(argv[3], "parts") && std::strcmp(argv[3], "cross")))) {
        std::fprintf(stderr, "usage: ref_reducer_probe <vmec.bin> <modes> parts|cross\n       ref_reducer_probe synthetic\n");
        return 2;
    }
    tables raw;
    size_t modes;
    if (synthetic) {
        raw.synthetic();
        modes = 2;
    } else {
        raw.read(argv[1]);
        modes = std::min<size_t> (std::strtoull(argv[2], nullptr, 10), raw.nummn);
    }
    const T sminh = raw.sminh, sminf = raw.sminf, ds = raw.ds;

    auto s = graph::variable<T> (1, "s");
    auto u = graph::variable<T> (1, "u");
    auto v = graph::variable<T> (1, "v");
    auto ks = graph::variable<T> (1, "k_s");
    auto ku = graph::variable<T> (1, "k_u");
    auto kv = graph::variable<T> (1, "k_v");
    auto w = graph::variable<T> (1, "\\omega");

//  vmec::set_cache, equilibrium.hpp:2083-2140
    auto s_norm_f = (s - sminf)/ds;
    auto zero = graph::zero<T> ();
    auto r = zero;
    auto z = zero;
    auto l = zero;
    for (size_t i = 0; i < modes; i++) {
        auto rmnc = build_1D_spline({graph::piecewise_1D(to_buffer(raw.rmnc[0][i]), s, ds, sminf), graph::piecewise_1D(to_buffer(raw.rmnc[1][i]), s, ds, sminf),
                                     graph::piecewise_1D(to_buffer(raw.rmnc[2][i]), s, ds, sminf), graph::piecewise_1D(to_buffer(raw.rmnc[3][i]), s, ds, sminf)},
                                    s, ds, sminf);
        auto zmns = build_1D_spline({graph::piecewise_1D(to_buffer(raw.zmns[0][i]), s, ds, sminf), graph::piecewise_1D(to_buffer(raw.zmns[1][i]), s, ds, sminf),
                                     graph::piecewise_1D(to_buffer(raw.zmns[2][i]), s, ds, sminf), graph::piecewise_1D(to_buffer(raw.zmns[3][i]), s, ds, sminf)},
                                    s, ds, sminf);
        auto lmns = build_1D_spline({graph::piecewise_1D(to_buffer(raw.lmns[0][i]), s, ds, sminh), graph::piecewise_1D(to_buffer(raw.lmns[1][i]), s, ds, sminh),
                                     graph::piecewise_1D(to_buffer(raw.lmns[2][i]), s, ds, sminh), graph::piecewise_1D(to_buffer(raw.lmns[3][i]), s, ds, sminh)},
                                    s, ds, sminh);
        auto m = graph::constant<T> (static_cast<T> (raw.xm[i]));
        auto n = graph::constant<T> (static_cast<T> (raw.xn[i]));
        auto sinmn = graph::sin(m*u - n*v);
        r = r + rmnc*graph::cos(m*u - n*v);
        z = z + zmns*sinmn;
        l = l + lmns*sinmn;
    }

//  get_esubs / get_esubu / get_esubv, equilibrium.hpp:1920-2013
    auto cosv = graph::cos(v);
    auto sinv = graph::sin(v);
    auto one = graph::one<T> ();
    auto rotation = graph::matrix(graph::vector(cosv, -sinv, zero),
                                  graph::vector(sinv, cosv,  zero),
                                  graph::vector(zero, zero,  one ));
    if (synthetic) {
        std::printf("cos(v)*dR/du with two modes ...\n");
        std::fflush(stdout);
        auto first = cosv*r->df(u);                    // the first product of rotation->dot(vector(dR/du, 0, dZ/du))
        std::printf("returned\n");
        (void)first;
        return 0;
    }
    auto esubs = rotation->dot(graph::vector(r->df(s), zero, z->df(s)));
    auto esubu = rotation->dot(graph::vector(r->df(u), zero, z->df(u)));
    auto esubv = rotation->dot(graph::vector(r->df(v), r,    z->df(v)));
    auto jacobian = esubs->dot(esubu->cross(esubv));
    auto esups = esubu->cross(esubv)/jacobian;
    auto esupu = esubv->cross(esubs)/jacobian;
    auto esupv = esubs->cross(esubu)/jacobian;
//  get_phi :2056-2059, get_chi :2036-2046 (evaluated at s_norm_f as :2133 does)
    auto phip = (graph::constant<T> (static_cast<T> (raw.signj))*graph::constant<T> (static_cast<T> (raw.dphi))*s)->df(s);
    auto chi = build_1D_spline({graph::piecewise_1D(to_buffer(raw.chi[0]), s_norm_f, ds, sminf), graph::piecewise_1D(to_buffer(raw.chi[1]), s_norm_f, ds, sminf),
                                graph::piecewise_1D(to_buffer(raw.chi[2]), s_norm_f, ds, sminf), graph::piecewise_1D(to_buffer(raw.chi[3]), s_norm_f, ds, sminf)},
                               s_norm_f, ds, sminf);
    auto jbsupu = chi->df(s) - phip*l->df(v);
    auto jbsupv = phip*(1.0 + l->df(u));
    auto bvec = (jbsupu*esubu + jbsupv*esubv)/jacobian;

//  dispersion_interface: k = k_s e^s + k_u e^u + k_v e^v (dispersion.hpp:1369-1434); cold_plasma::D :995-1001.
    auto k = ks*esups + ku*esupu + kv*esupv;
    auto n = k/w;
    auto b_hat = bvec->unit();

    if (!std::strcmp(argv[3], "parts")) {
        timed("B_x", bvec->get_x(), s);
        timed("|B|", bvec->length(), s);
        timed("b_x", b_hat->get_x(), s);
        timed("k_x", k->get_x(), s);
        timed("n.b", b_hat->dot(n), s);
        timed("n.n", n->dot(n), s);
        std::printf("parts done\n");
        return 0;
    }
    std::printf("differentiating (b x n)_x with respect to s ...\n");
    std::fflush(stdout);
    timed("(bxn)_x", b_hat->cross(n)->get_x(), s);
    std::printf("cross returned\n");
    return 0;
}

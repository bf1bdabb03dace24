// ---------------------------------------------------------------------------
// ref_builders.hpp — shared by oracle/ref_driver.cpp and oracle/hip_context_demo.cpp
// (TEST INFRASTRUCTURE; builds only where /root/reference is present).
//
// Contents: a tape interpreter for reference DAGs (one IEEE operation per node, as the
// node's compile() emits it), and the restatement — against the reference's own node API —
// of the graph construction that equilibrium.hpp / dispersion.hpp / solver.hpp / newton.hpp
// perform for the hot path (those headers cannot be included: NetCDF-C, LLVM JIT).
// ---------------------------------------------------------------------------
#ifndef ref_builders_hpp
#define ref_builders_hpp

#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <limits>
#include <iostream>
#include <algorithm>
#include <functional>

#include "node.hpp"
#include "arithmetic.hpp"
#include "math.hpp"
#include "trigonometry.hpp"
#include "piecewise.hpp"
#include "vector.hpp"
#include "random.hpp"
#include "special_functions.hpp"

#include "../graph_framework_amd/gfir_serialize.hpp"

template<typename T, bool S=false> using leaf = graph::shared_leaf<T, S>;
template<typename T, bool S=false> using vec3 = graph::shared_vector<T, S>;

// ---------------------------------------------------------------------------
// Tape interpreter.
// ---------------------------------------------------------------------------
enum class op_t {constant, input, add, sub, mul, div, fma, sqrt, powi, pow,
                 sin, cos, atan2, exp, log, gather1, gather2, erfi};

template<typename T>
struct instruction {
    op_t op;
    int a = -1, b = -1, c = -1;
    T value = 0;            // constant value
    size_t power = 0;       // integer exponent
    const T *table = nullptr;
    size_t length = 0, num_columns = 0, num_rows = 0;
    T scale = 1, offset = 0, y_scale = 1, y_offset = 0;
};

//  T real: one IEEE operation per node.  T complex: the host's std::complex operators and functions, which
//  is what the kernels cpu_context compiles are written in (jit::add_type, register.hpp).  S: the
//  SAFE_MATH guards the nodes emit (arithmetic.hpp:2534-2557, :3526-3541, :5101-5117, math.hpp:450-471).
template<typename T, bool S=false>
class tape {
public:
    std::vector<instruction<T>> code;
    std::map<graph::leaf_node<T, S> *, int> slots;
    std::map<graph::leaf_node<T, S> *, int> inputs;        // variable node -> input column
    std::vector<std::vector<T>> tables;                 // owned copies of gather tables
    std::map<op_t, size_t> counts;

    int emit(const instruction<T> &i) {
        code.push_back(i);
        counts[i.op]++;
        return static_cast<int> (code.size() - 1);
    }

    const T *own_table(const backend::buffer<T> &b) {
        std::vector<T> t(b.size());
        for (size_t i = 0; i < b.size(); i++) t[i] = b[i];
        tables.push_back(std::move(t));
        return tables.back().data();
    }

    int lower_erfi(leaf<T, S> n) {                              // math.hpp:1440 ff., complex base types only
        if constexpr (jit::complex_scalar<T>) {
            if (auto x = graph::erfi_cast(n); x.get()) {
                instruction<T> ins;
                ins.op = op_t::erfi;
                ins.a = lower(x->get_arg());
                return emit(ins);
            }
        }
        return -1;
    }

//  Lower a node in the order compile() recurses (left, [middle,] right, self).
    int lower(leaf<T, S> n) {
        auto found = slots.find(n.get());
        if (found != slots.end()) {
            return found->second;
        }
        instruction<T> ins;
        int slot = -1;
        if (auto c = graph::constant_cast(n); c.get()) {
            ins.op = op_t::constant;
            ins.value = c->evaluate().at(0);                    // node.hpp:793-808
            slot = emit(ins);
        } else if (auto v = graph::variable_cast(n); v.get()) {
            auto in = inputs.find(n.get());
            if (in == inputs.end()) {
                std::cerr << "unbound variable " << std::endl;
                exit(1);
            }
            ins.op = op_t::input;
            ins.a = in->second;
            slot = emit(ins);
        } else if (auto p = graph::pseudo_variable_cast(n); p.get()) {
            slot = lower(p->get_arg());                         // node.hpp:1745 ff.
        } else if (auto x = graph::add_cast(n); x.get()) {
            ins.op = op_t::add;                                 // arithmetic.hpp:645-669
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            slot = emit(ins);
        } else if (auto x = graph::subtract_cast(n); x.get()) {
            ins.op = op_t::sub;                                 // arithmetic.hpp:1475 ff.
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            slot = emit(ins);
        } else if (auto x = graph::multiply_cast(n); x.get()) {
            ins.op = op_t::mul;                                 // arithmetic.hpp:2516 ff.
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            slot = emit(ins);
        } else if (auto x = graph::divide_cast(n); x.get()) {
            ins.op = op_t::div;                                 // arithmetic.hpp:3508 ff.
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            slot = emit(ins);
        } else if (auto x = graph::fma_cast(n); x.get()) {
            ins.op = op_t::fma;                                 // arithmetic.hpp:5079-5127
            ins.a = lower(x->get_left()); ins.b = lower(x->get_middle());
            ins.c = lower(x->get_right());
            slot = emit(ins);
        } else if (auto x = graph::sqrt_cast(n); x.get()) {
            ins.op = op_t::sqrt;                                // math.hpp:166 ff.
            ins.a = lower(x->get_arg());
            slot = emit(ins);
        } else if (auto x = graph::pow_cast(n); x.get()) {
            ins.a = lower(x->get_left());                       // math.hpp:1199-1230
            auto e = graph::constant_cast(x->get_right());
            if (e.get() && e->is_integer()) {
                ins.op = op_t::powi;
                ins.power = static_cast<size_t> (std::real(x->get_right()->evaluate().at(0)));
            } else {
                ins.op = op_t::pow;
                ins.b = lower(x->get_right());
            }
            slot = emit(ins);
        } else if (auto x = graph::exp_cast(n); x.get()) {
            ins.op = op_t::exp; ins.a = lower(x->get_arg()); slot = emit(ins);
        } else if (auto x = graph::log_cast(n); x.get()) {
            ins.op = op_t::log; ins.a = lower(x->get_arg()); slot = emit(ins);
        } else if (auto x = graph::sin_cast(n); x.get()) {
            ins.op = op_t::sin; ins.a = lower(x->get_arg()); slot = emit(ins);
        } else if (auto x = graph::cos_cast(n); x.get()) {
            ins.op = op_t::cos; ins.a = lower(x->get_arg()); slot = emit(ins);
        } else if (auto x = graph::atan_cast(n); x.get()) {
            ins.op = op_t::atan2;                               // trigonometry.hpp:718: atan2(r, l)
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            slot = emit(ins);
        } else if (const int e = lower_erfi(n); e >= 0) {
            slot = e;
        } else if (auto x = graph::piecewise_1D_cast(n); x.get()) {
            ins.op = op_t::gather1;                             // piecewise.hpp:349-437
            ins.a = lower(x->get_arg());
            const backend::buffer<T> data = x->evaluate();
            ins.table = own_table(data);
            ins.length = data.size();
            ins.scale = x->get_scale();
            ins.offset = x->get_offset();
            slot = emit(ins);
        } else if (auto x = graph::piecewise_2D_cast(n); x.get()) {
            ins.op = op_t::gather2;                             // piecewise.hpp:1072-1208
            ins.a = lower(x->get_left()); ins.b = lower(x->get_right());
            const backend::buffer<T> data = x->evaluate();
            ins.table = own_table(data);
            ins.length = data.size();
            ins.num_columns = x->get_num_columns();
            ins.num_rows = x->get_num_rows();
            ins.scale = x->get_x_scale(); ins.offset = x->get_x_offset();
            ins.y_scale = x->get_y_scale(); ins.y_offset = x->get_y_offset();
            slot = emit(ins);
        } else {
            std::cerr << "unsupported node type in tape lowering" << std::endl;
            exit(1);
        }
        slots[n.get()] = slot;
        return slot;
    }

//  compile_index, piecewise.hpp:26-65.
    static size_t index(const T x, const T scale, const T offset, const size_t length) {
        const auto q = std::real((x - offset)/scale);                       // complex: real(...) :43-47
        typedef decltype(q) real;
        return static_cast<size_t> (std::min<real> (std::max<real> (q, 0), static_cast<real> (length - 1)));
    }

    static T multiply(const T a, const T b) {
        if constexpr (S) {
            if (a == static_cast<T> (0) || b == static_cast<T> (0)) return static_cast<T> (0);
        }
        return a*b;
    }

    void run(const T *in, std::vector<T> &r) const {
        r.resize(code.size());
        for (size_t i = 0, ie = code.size(); i < ie; i++) {
            const instruction<T> &c = code[i];
            switch (c.op) {
                case op_t::constant: r[i] = c.value; break;
                case op_t::input:    r[i] = in[c.a]; break;
                case op_t::add:      r[i] = r[c.a] + r[c.b]; break;
                case op_t::sub:      r[i] = r[c.a] - r[c.b]; break;
                case op_t::mul:      r[i] = multiply(r[c.a], r[c.b]); break;
                case op_t::div:
                    if constexpr (S) {
                        if (r[c.a] == static_cast<T> (0)) { r[i] = 0; break; }
                    }
                    r[i] = r[c.a]/r[c.b];
                    break;
                case op_t::fma:
                    if constexpr (S) {
                        if (r[c.a] == static_cast<T> (0) || r[c.b] == static_cast<T> (0)) { r[i] = r[c.c]; break; }
                    }
                    if constexpr (jit::complex_scalar<T>) {
                        r[i] = r[c.a]*r[c.b] + r[c.c];
                    } else {
                        r[i] = std::fma(r[c.a], r[c.b], r[c.c]);
                    }
                    break;
                case op_t::sqrt:     r[i] = std::sqrt(r[c.a]); break;
                case op_t::powi: {
                    T v = r[c.a];
                    for (size_t k = 1; k < c.power; k++) v = v*r[c.a];
                    r[i] = v;
                    break;
                }
                case op_t::pow:      r[i] = std::pow(r[c.a], r[c.b]); break;
                case op_t::sin:      r[i] = std::sin(r[c.a]); break;
                case op_t::cos:      r[i] = std::cos(r[c.a]); break;
                case op_t::atan2:
                    if constexpr (jit::complex_scalar<T>) {
                        r[i] = std::atan(r[c.b]/r[c.a]);                    // trigonometry.hpp:712-716
                    } else {
                        r[i] = std::atan2(r[c.b], r[c.a]);
                    }
                    break;
                case op_t::exp:
                    if constexpr (S) {
                        if (!(std::real(r[c.a]) < 709.8)) { r[i] = static_cast<T> (std::numeric_limits<decltype(std::real(T()))>::max()); break; }
                    }
                    r[i] = std::exp(r[c.a]);
                    break;
                case op_t::log:      r[i] = std::log(r[c.a]); break;
                case op_t::erfi:
                    if constexpr (jit::complex_scalar<T>) {
                        r[i] = special::erfi(r[c.a]);                       // special_functions.hpp:1583
                    }
                    break;
                case op_t::gather1:
                    r[i] = c.table[index(r[c.a], c.scale, c.offset, c.length)];
                    break;
                case op_t::gather2:
                    r[i] = c.table[index(r[c.a], c.scale, c.offset, c.num_rows)*c.num_columns +
                                   index(r[c.b], c.y_scale, c.y_offset, c.num_columns)];
                    break;
            }
        }
    }

    void print_counts(FILE *f) const {
        static const char *names[] = {"constant", "input", "add", "sub", "mul", "div", "fma", "sqrt",
                                      "powi", "pow", "sin", "cos", "atan2", "exp", "log", "gather1", "gather2", "erfi"};
        fprintf(f, "{\"statements\": %zu", code.size());
        for (auto &kv : counts) fprintf(f, ", \"%s\": %zu", names[static_cast<int> (kv.first)], kv.second);
        fprintf(f, "}\n");
    }
};

//  A work item (workflow.hpp:22-76): inputs, outputs, setters(expression -> variable).
template<typename T, bool S=false>
struct work_item {
    tape<T, S> code;
    std::vector<int> output_slots;
    std::vector<std::pair<int, int>> setter_slots;      // (expression slot, input column)
    std::vector<leaf<T, S>> in_nodes, out_nodes;
    std::vector<std::pair<leaf<T, S>, leaf<T, S>>> set_nodes;

//  The same item as GFIR bytes (what hip_context hands to the HIP backend).
    void write_gfir(const std::string &name, const std::string &path) const {
        graph::input_nodes<T, S> in;
        for (auto &i : in_nodes) in.push_back(graph::variable_cast(i));
        graph::map_nodes<T, S> set;
        for (auto &s : set_nodes) set.push_back({s.first, graph::variable_cast(s.second)});
        gfir::serializer<T, S> ser;
        const std::vector<uint8_t> bytes = ser(name, in, out_nodes, set);
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) { perror(path.c_str()); exit(1); }
        fwrite(bytes.data(), 1, bytes.size(), f);
        fclose(f);
        fprintf(stderr, "wrote %s (%zu bytes)\n", path.c_str(), bytes.size());
    }

    work_item(const std::vector<leaf<T, S>> &inputs, const std::vector<leaf<T, S>> &outputs,
              const std::vector<std::pair<leaf<T, S>, leaf<T, S>>> &setters) :
    in_nodes(inputs), out_nodes(outputs), set_nodes(setters) {
        for (size_t i = 0; i < inputs.size(); i++) {
            code.inputs[inputs[i].get()] = static_cast<int> (i);
        }
//  jit::context::add_kernel compiles setters first, then outputs (jit.hpp:170-178).
        for (auto &s : setters) {
            const int slot = code.lower(s.first);
            setter_slots.push_back({slot, code.inputs.at(s.second.get())});
        }
        for (auto &o : outputs) {
            output_slots.push_back(code.lower(o));
        }
    }

//  Run over n elements of SoA columns; writes outputs, then applies setters in place.
    void run(const size_t n, std::vector<T *> columns, std::vector<T *> outs) const {
        std::vector<T> in(columns.size()), regs;
        for (size_t i = 0; i < n; i++) {
            for (size_t c = 0; c < columns.size(); c++) in[c] = columns[c][i];
            code.run(in.data(), regs);
            for (size_t o = 0; o < output_slots.size(); o++) outs[o][i] = regs[output_slots[o]];
            for (auto &s : setter_slots) columns[s.second][i] = regs[s.first];
        }
    }
};

// ---------------------------------------------------------------------------
// EFIT tables file written by tests/golden/make_ref_golden.py:
//   9 doubles (rmin dr zmin dz psimin dpsi ne_scale te_scale pres_scale),
//   3 uint64 (numr numz numpsi), 16 psi tables, then te, ne, pressure, fpol (4 each).
// ---------------------------------------------------------------------------
struct raw_tables {
    double scalars[9];
    uint64_t numr, numz, numpsi;
    std::vector<double> psi[16], te[4], ne[4], pres[4], fpol[4];

    explicit raw_tables(const char *path) {
        FILE *f = fopen(path, "rb");
        if (!f) { perror(path); exit(1); }
        auto rd = [f] (void *p, size_t bytes) { if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(1); } };
        rd(scalars, sizeof(scalars));
        rd(&numr, 8); rd(&numz, 8); rd(&numpsi, 8);
        for (auto &t : psi) { t.resize(numr*numz); rd(t.data(), 8*t.size()); }
        for (auto *g : {te, ne, pres, fpol}) {
            for (int k = 0; k < 4; k++) { g[k].resize(numpsi); rd(g[k].data(), 8*numpsi); }
        }
        fclose(f);
    }
};

// ---------------------------------------------------------------------------
// equilibrium::generic (equilibrium.hpp:236-470): what the dispersion relations ask of
// an equilibrium.  One ion species everywhere on the path (deuterium mass 3.34449469e-27,
// charge 1: equilibrium.hpp:489,618,742,871,998,1475).
// ---------------------------------------------------------------------------
template<typename T, bool S=false>
struct equilibrium_base {
    virtual ~equilibrium_base() {}
    virtual leaf<T, S> get_electron_density(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) = 0;
    virtual leaf<T, S> get_ion_density(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) = 0;
    virtual leaf<T, S> get_electron_temperature(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) = 0;
    virtual leaf<T, S> get_ion_temperature(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) = 0;
    virtual vec3<T, S> get_magnetic_field(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) = 0;

//  generic::get_esup1..3, equilibrium.hpp:379-420.
    vec3<T, S> esup(const int i) {
        auto one = graph::one<T, S> ();
        auto zero = graph::zero<T, S> ();
        return i == 0 ? graph::vector(one, zero, zero)
             : (i == 1 ? graph::vector(zero, one, zero) : graph::vector(zero, zero, one));
    }
//  ... as functions of the coordinates, for equilibria in generalised coordinates (vmec).
    virtual vec3<T, S> esup_at(const int i, leaf<T, S>, leaf<T, S>, leaf<T, S>) { return esup(i); }
};

// ---------------------------------------------------------------------------
// Restatement of equilibrium::efit (equilibrium.hpp:1146-1616) against the
// reference node API.  Member-init bug of :1478 and ni = te of :1361 kept.
// ---------------------------------------------------------------------------
template<typename T, bool S=false>
class efit : public equilibrium_base<T, S> {
public:
    T psimin, dpsi, rmin, dr, zmin, dz;
    backend::buffer<T> te_c[4], ne_c[4], pres_c[4], fpol_c[4], c[4][4];
    leaf<T, S> te_scale, ne_scale, pres_scale;
    size_t num_cols;

    leaf<T, S> x_cache, y_cache, z_cache, ne_cache, ni_cache, te_cache, ti_cache, psi_cache;
    vec3<T, S> b_cache;

    static backend::buffer<T> to_buffer(const std::vector<double> &v) {     // :1807-1842
        return backend::buffer<T> (std::vector<T> (v.begin(), v.end()));
    }

    explicit efit(const raw_tables &raw) {                                  // make_efit :1797-1853
        rmin = static_cast<T> (raw.scalars[0]); dr = static_cast<T> (raw.scalars[1]);
        zmin = static_cast<T> (raw.scalars[2]); dz = static_cast<T> (raw.scalars[3]);
        psimin = static_cast<T> (raw.scalars[4]); dpsi = static_cast<T> (raw.scalars[5]);
        ne_scale = graph::constant<T, S> (static_cast<T> (raw.scalars[6]));
        te_scale = graph::constant<T, S> (static_cast<T> (raw.scalars[7]));
        pres_scale = graph::constant<T, S> (static_cast<T> (raw.scalars[8]));
        num_cols = raw.numz;                                                // :1849
        for (int a = 0; a < 4; a++) {
            for (int b = 0; b < 4; b++) c[a][b] = to_buffer(raw.psi[a*4 + b]);
            te_c[a] = to_buffer(raw.te[a]);
            ne_c[a] = to_buffer(raw.ne[a]);
            pres_c[a] = to_buffer(raw.pres[a]);
            fpol_c[a] = to_buffer(raw.fpol[a]);
        }
        ne_c[0] = te_c[0];                                                  // :1478
        ne_c[1] = te_c[1];
        auto zero = graph::zero<T, S> ();                                      // :1486-1489
        x_cache = zero; y_cache = zero; z_cache = zero;
    }

    static leaf<T, S> build_1D_spline(std::vector<leaf<T, S>> cc, leaf<T, S> x,     // :1121-1131
                                   const T scale, const T offset) {
        auto c3 = cc[3]/(scale*scale*scale);
        auto c2 = cc[2]/(scale*scale) - static_cast<T> (3.0)*offset*cc[3]/(scale*scale*scale);
        auto c1 = cc[1]/scale - static_cast<T> (2.0)*offset*cc[2]/(scale*scale) + static_cast<T> (3.0)*offset*offset*cc[3]/(scale*scale*scale);
        auto c0 = cc[0] - offset*cc[1]/scale + offset*offset*cc[2]/(scale*scale) - offset*offset*offset*cc[3]/(scale*scale*scale);
        return graph::fma(graph::fma(graph::fma(c3, x, c2), x, c1), x, c0);
    }

    leaf<T, S> build_psi(leaf<T, S> r, const T r_scale, const T r_offset,        // :1279-1313
                      leaf<T, S> z, const T z_scale, const T z_offset) {
        leaf<T, S> t[4][4];
        for (int a = 0; a < 4; a++) {
            for (int b = 0; b < 4; b++) {
                t[a][b] = graph::piecewise_2D(c[a][b], num_cols, r, r_scale, r_offset, z, z_scale, z_offset);
            }
        }
        auto r_norm = (r - r_offset)/r_scale;
        auto c0 = build_1D_spline({t[0][0], t[0][1], t[0][2], t[0][3]}, z, z_scale, z_offset);
        auto c1 = build_1D_spline({t[1][0], t[1][1], t[1][2], t[1][3]}, z, z_scale, z_offset);
        auto c2 = build_1D_spline({t[2][0], t[2][1], t[2][2], t[2][3]}, z, z_scale, z_offset);
        auto c3 = build_1D_spline({t[3][0], t[3][1], t[3][2], t[3][3]}, z, z_scale, z_offset);
        return ((c3*r_norm + c2)*r_norm + c1)*r_norm + c0;
    }

    leaf<T, S> profile(const backend::buffer<T> cc[4]) {
        auto p0 = graph::piecewise_1D(cc[0], psi_cache, dpsi, psimin);
        auto p1 = graph::piecewise_1D(cc[1], psi_cache, dpsi, psimin);
        auto p2 = graph::piecewise_1D(cc[2], psi_cache, dpsi, psimin);
        auto p3 = graph::piecewise_1D(cc[3], psi_cache, dpsi, psimin);
        return build_1D_spline({p0, p1, p2, p3}, psi_cache, dpsi, psimin);
    }

    void set_cache(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) {                      // :1324-1384
        if (!x->is_match(x_cache) || !y->is_match(y_cache) || !z->is_match(z_cache)) {
            x_cache = x; y_cache = y; z_cache = z;

            auto r = graph::sqrt(x*x + y*y);
            psi_cache = build_psi(r, dr, rmin, z, dz, zmin);

            ne_cache = ne_scale*profile(ne_c);
            te_cache = te_scale*profile(te_c);
            auto pressure = pres_scale*profile(pres_c);

            auto q = graph::constant<T, S> (static_cast<T> (1.60218E-19));
            ni_cache = te_cache;                                            // :1361
            ti_cache = (pressure - ne_cache*te_cache*q)/(ni_cache*q);

            auto phi = graph::atan(x, y);
            auto br = psi_cache->df(z)/r;
            auto bp = profile(fpol_c)/r;
            auto bz = -psi_cache->df(r)/r;
            auto cos = graph::cos(phi);
            auto sin = graph::sin(phi);
            b_cache = graph::vector(br*cos - bp*sin, br*sin + bp*cos, bz);
        }
    }

    leaf<T, S> get_electron_density(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) override { set_cache(x, y, z); return ne_cache; }
    leaf<T, S> get_ion_density(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) override { set_cache(x, y, z); return ni_cache; }
    leaf<T, S> get_electron_temperature(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) override { set_cache(x, y, z); return te_cache; }
    leaf<T, S> get_ion_temperature(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) override { set_cache(x, y, z); return ti_cache; }
    vec3<T, S> get_magnetic_field(leaf<T, S> x, leaf<T, S> y, leaf<T, S> z) override { set_cache(x, y, z); return b_cache; }
};

// ---------------------------------------------------------------------------
// VMEC tables file written by tests/golden/make_vmec_golden.py: 5 doubles (sminh sminf ds dphi signj),
// 3 uint64 (numsf numsh nummn), chi_c0..3 (numsf each), then per quantity and coefficient the nummn
// rows: rmnc_c0..3, zmns_c0..3 (numsf per row), lmns_c0..3 (numsh per row), xm, xn.
// ---------------------------------------------------------------------------
struct raw_vmec {
    double scalars[5];
    uint64_t numsf, numsh, nummn;
    std::vector<double> chi[4], xm, xn;
    std::vector<std::vector<double>> rmnc[4], zmns[4], lmns[4];

    explicit raw_vmec(const char *path) {
        FILE *f = fopen(path, "rb");
        if (!f) { perror(path); exit(1); }
        auto rd = [f] (void *p, size_t bytes) { if (fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(1); } };
        rd(scalars, sizeof(scalars));
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
        fclose(f);
    }
};

// ---------------------------------------------------------------------------
// Restatement of equilibrium::vmec (equilibrium.hpp:1868-2330) against the reference node API:
// flux coordinates (s, u, v), R, Z, lambda as Fourier sums over `nummn` modes of cubic splines in s,
// covariant basis from df(), contravariant basis and B from the Jacobian.
// ---------------------------------------------------------------------------
template<typename T, bool S=false>
class vmec : public equilibrium_base<T, S> {
public:
    T sminh, sminf, ds;
    leaf<T, S> signj, dphi;
    backend::buffer<T> chi_c[4];
    std::vector<backend::buffer<T>> rmnc_c[4], zmns_c[4], lmns_c[4];
    std::vector<T> xm, xn;
    size_t modes;

    leaf<T, S> s_cache, u_cache, v_cache, x_cache, y_cache, z_cache;
    vec3<T, S> esups_cache, esupu_cache, esupv_cache, bvec_cache;

    static backend::buffer<T> to_buffer(const std::vector<double> &v) {
        return backend::buffer<T> (std::vector<T> (v.begin(), v.end()));
    }

    explicit vmec(const raw_vmec &raw, const size_t use_modes = 0) {        // make_vmec :2332-2650
        sminh = static_cast<T> (raw.scalars[0]);
        sminf = static_cast<T> (raw.scalars[1]);
        ds = static_cast<T> (raw.scalars[2]);
        dphi = graph::constant<T, S> (static_cast<T> (raw.scalars[3]));
        signj = graph::constant<T, S> (static_cast<T> (raw.scalars[4]));
        modes = use_modes ? std::min<size_t> (use_modes, raw.nummn) : raw.nummn;
        for (int k = 0; k < 4; k++) {
            chi_c[k] = to_buffer(raw.chi[k]);
            for (size_t i = 0; i < modes; i++) {
                rmnc_c[k].push_back(to_buffer(raw.rmnc[k][i]));
                zmns_c[k].push_back(to_buffer(raw.zmns[k][i]));
                lmns_c[k].push_back(to_buffer(raw.lmns[k][i]));
            }
        }
        xm.assign(raw.xm.begin(), raw.xm.end());
        xn.assign(raw.xn.begin(), raw.xn.end());
        auto zero = graph::zero<T, S> ();                                   // :2212-2215
        s_cache = zero; u_cache = zero; v_cache = zero;
    }

    static leaf<T, S> spline(std::vector<leaf<T, S>> cc, leaf<T, S> x, const T scale, const T offset) {
        return efit<T, S>::build_1D_spline(cc, x, scale, offset);           // equilibrium.hpp:1121-1131
    }

    vec3<T, S> rotate(vec3<T, S> column) {                                  // get_esubs/u/v :1920-1975
        auto cosv = graph::cos(v_cache);
        auto sinv = graph::sin(v_cache);
        auto one = graph::one<T, S> ();
        auto zero = graph::zero<T, S> ();
        auto m = graph::matrix(graph::vector(cosv, -sinv, zero),
                               graph::vector(sinv, cosv,  zero),
                               graph::vector(zero, zero,  one ));
        return m->dot(column);
    }

    leaf<T, S> get_chi(leaf<T, S> s) {                                      // :1984-1992
        auto c0 = graph::piecewise_1D(chi_c[0], s, ds, sminf);
        auto c1 = graph::piecewise_1D(chi_c[1], s, ds, sminf);
        auto c2 = graph::piecewise_1D(chi_c[2], s, ds, sminf);
        auto c3 = graph::piecewise_1D(chi_c[3], s, ds, sminf);
        return spline({c0, c1, c2, c3}, s, ds, sminf);
    }

    void set_cache(leaf<T, S> s, leaf<T, S> u, leaf<T, S> v) {              // :1999-2074
        if (!s->is_match(s_cache) || !u->is_match(u_cache) || !v->is_match(v_cache)) {
            s_cache = s; u_cache = u; v_cache = v;

            auto s_norm_f = (s - sminf)/ds;

            auto zero = graph::zero<T, S> ();
            auto r = zero;
            auto z = zero;
            auto l = zero;

            for (size_t i = 0; i < modes; i++) {
                auto rmnc = spline({graph::piecewise_1D(rmnc_c[0][i], s, ds, sminf), graph::piecewise_1D(rmnc_c[1][i], s, ds, sminf),
                                    graph::piecewise_1D(rmnc_c[2][i], s, ds, sminf), graph::piecewise_1D(rmnc_c[3][i], s, ds, sminf)},
                                   s, ds, sminf);
                auto zmns = spline({graph::piecewise_1D(zmns_c[0][i], s, ds, sminf), graph::piecewise_1D(zmns_c[1][i], s, ds, sminf),
                                    graph::piecewise_1D(zmns_c[2][i], s, ds, sminf), graph::piecewise_1D(zmns_c[3][i], s, ds, sminf)},
                                   s, ds, sminf);
                auto lmns = spline({graph::piecewise_1D(lmns_c[0][i], s, ds, sminh), graph::piecewise_1D(lmns_c[1][i], s, ds, sminh),
                                    graph::piecewise_1D(lmns_c[2][i], s, ds, sminh), graph::piecewise_1D(lmns_c[3][i], s, ds, sminh)},
                                   s, ds, sminh);

                auto m = graph::constant<T, S> (xm[i]);
                auto n = graph::constant<T, S> (xn[i]);

                auto sinmn = graph::sin(m*u - n*v);

                r = r + rmnc*graph::cos(m*u - n*v);
                z = z + zmns*sinmn;
                l = l + lmns*sinmn;
            }

            x_cache = r*graph::cos(v);
            y_cache = r*graph::sin(v);
            z_cache = z;

            auto esubs = rotate(graph::vector(r->df(s_cache), zero, z->df(s_cache)));
            auto esubu = rotate(graph::vector(r->df(u_cache), zero, z->df(u_cache)));
            auto esubv = rotate(graph::vector(r->df(v_cache), r,    z->df(v_cache)));

            auto jacobian = esubs->dot(esubu->cross(esubv));

            esups_cache = esubu->cross(esubv)/jacobian;
            esupu_cache = esubv->cross(esubs)/jacobian;
            esupv_cache = esubs->cross(esubu)/jacobian;

            auto phip = (signj*dphi*s)->df(s);
            auto jbsupu = get_chi(s_norm_f)->df(s) - phip*l->df(v);
            auto jbsupv = phip*(1.0 + l->df(u));
            bvec_cache = (jbsupu*esubu + jbsupv*esubv)/jacobian;
        }
    }

    leaf<T, S> get_profile(leaf<T, S> s) {                                  // :2076-2079
        return graph::pow((1.0 - graph::pow(graph::sqrt(s*s), 1.5)), 2.0);
    }

    vec3<T, S> esup_at(const int i, leaf<T, S> s, leaf<T, S> u, leaf<T, S> v) override {
        set_cache(s, u, v);
        return i == 0 ? esups_cache : (i == 1 ? esupu_cache : esupv_cache);
    }
    leaf<T, S> get_electron_density(leaf<T, S> s, leaf<T, S>, leaf<T, S>) override {
        return graph::constant<T, S> (static_cast<T> (1.0E19))*get_profile(s);
    }
    leaf<T, S> get_ion_density(leaf<T, S> s, leaf<T, S> u, leaf<T, S> v) override { return get_electron_density(s, u, v); }
    leaf<T, S> get_electron_temperature(leaf<T, S> s, leaf<T, S>, leaf<T, S>) override {
        return graph::constant<T, S> (static_cast<T> (1000.0))*get_profile(s);
    }
    leaf<T, S> get_ion_temperature(leaf<T, S> s, leaf<T, S> u, leaf<T, S> v) override { return get_electron_temperature(s, u, v); }
    vec3<T, S> get_magnetic_field(leaf<T, S> s, leaf<T, S> u, leaf<T, S> v) override { set_cache(s, u, v); return bvec_cache; }
};

// ---------------------------------------------------------------------------
// dispersion::cold_plasma::D, dispersion.hpp:941-1008 (constants :490-503,
// helpers :326-332, :348-353; ion species equilibrium.hpp:1475).
// ---------------------------------------------------------------------------
template<typename T>
leaf<T> cold_plasma_D(leaf<T> w, vec3<T> k_vec, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq,
                      std::vector<leaf<T>> *debug = nullptr) {
    const T epsilon0 = 8.8541878138E-12;
    const T mu0 = M_PI*4.0E-7;
    const T q = 1.602176634E-19;
    const T me = 9.1093837015E-31;
    const T c = static_cast<T> (1.0)/std::sqrt(epsilon0*mu0);
    auto plasma_frequency = [] (leaf<T> n, const T q_, const T m, const T c_, const T eps) {
        return n*q_*q_/(eps*m*c_*c_);
    };
    auto cyclotron_frequency = [] (const T q_, leaf<T> b, const T m, const T c_) {
        return q_*b/(m*c_);
    };

    auto ne = eq.get_electron_density(x, y, z);
    auto wpe2 = plasma_frequency(ne, q, me, c, epsilon0);
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto b_len = b_vec->length();
    auto ec = cyclotron_frequency(-q, b_len, me, c);

    auto w2 = w*w;
    auto denome = 1.0 - ec*ec/w2;
    auto e11 = 1.0 - (wpe2/w2)/denome;
    auto e12 = ((ec/w)*(wpe2/w2))/denome;
    auto e33 = wpe2;

    {
        const T mi = 3.34449469E-27;
        const T charge = static_cast<T> (static_cast<uint8_t> (1))*q;
        auto ni = eq.get_ion_density(x, y, z);
        auto wpi2 = plasma_frequency(ni, charge, mi, c, epsilon0);
        auto ic = cyclotron_frequency(charge, b_len, mi, c);
        auto denomi = 1.0 - ic*ic/w2;
        e11 = e11 - (wpi2/w2)/denomi;
        e12 = e12 + ((ic/w)*(wpi2/w2))/denomi;
        e33 = e33 + wpi2;
    }

    e12 = -1.0*e12;
    e33 = 1.0 - e33/w2;

    auto n = k_vec/w;
    auto b_hat = b_vec->unit();
    auto npara = b_hat->dot(n);
    auto npara2 = npara*npara;
    auto nperp = b_hat->cross(n)->length();
    auto nperp2 = nperp*nperp;

    auto m11 = e11 - npara2;
    auto m12 = e12;
    auto m13 = npara*nperp;
    auto m22 = e11 - npara2 - nperp2;
    auto m33 = e33 - nperp2;

    if (debug) {
        auto cr = b_hat->cross(n);
        *debug = {e11, e12, e33, npara, nperp, nperp2, m11, m13, m22, m33, b_len,
                  b_hat->get_x(), b_hat->get_y(), b_hat->get_z(), n->get_x(), n->get_y(), n->get_z(),
                  cr->get_x(), cr->get_y(), cr->get_z(), cr->dot(cr),
                  cr->get_x()*cr->get_x(), cr->get_y()*cr->get_y(), cr->get_z()*cr->get_z(), npara2};
    }
    return (m11*m22 - m12*m12)*m33 - m22*(m13*m13);
}

// ---------------------------------------------------------------------------
// dispersion::ordinary_wave::D, dispersion.hpp:785-812:  D = 1 - wpe^2/w^2 - nperp^2.
// ---------------------------------------------------------------------------
template<typename T>
leaf<T> ordinary_wave_D(leaf<T> w, vec3<T> k_vec, leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq,
                        std::vector<leaf<T>> *debug = nullptr) {
    (void)debug;
    const T epsilon0 = 8.8541878138E-12;
    const T mu0 = M_PI*4.0E-7;
    const T q = 1.602176634E-19;
    const T me = 9.1093837015E-31;
    const T c = static_cast<T> (1.0)/std::sqrt(epsilon0*mu0);
    auto ne = eq.get_electron_density(x, y, z);
    auto wpe2 = ne*q*q/(epsilon0*me*c*c);
    auto n = k_vec/w;
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto b_hat = b_vec->unit();
    auto nperp = b_hat->cross(n);
    auto nperp2 = nperp->dot(nperp);
    auto w2 = w*w;
    return 1.0 - wpe2/w2 - nperp2;
}

template<typename T>
using dispersion_function = leaf<T> (*)(leaf<T>, vec3<T>, leaf<T>, leaf<T>, leaf<T>, equilibrium_base<T> &,
                                        std::vector<leaf<T>> *);

// ---------------------------------------------------------------------------
// dispersion::dispersion_interface ctor, dispersion.hpp:1369-1434.
// ---------------------------------------------------------------------------
template<typename T>
struct dispersion_interface {
    vec3<T> k_vec;
    leaf<T> D, dxdt, dydt, dzdt, dkxdt, dkydt, dkzdt;
    leaf<T> dDdw, dDdkx, dDdky, dDdkz, dDdx, dDdy, dDdz;

    dispersion_function<T> function;

    dispersion_interface(leaf<T> w, leaf<T> kx, leaf<T> ky, leaf<T> kz,
                         leaf<T> x, leaf<T> y, leaf<T> z, equilibrium_base<T> &eq,
                         dispersion_function<T> f = cold_plasma_D<T>) :
    k_vec(kx*eq.esup_at(0, x, y, z) + ky*eq.esup_at(1, x, y, z) + kz*eq.esup_at(2, x, y, z)),
    D(f(w, k_vec, x, y, z, eq, nullptr)), function(f) {
        auto dkdx = k_vec->df(x);
        auto dkdy = k_vec->df(y);
        auto dkdz = k_vec->df(z);
        auto dDdk_vec = graph::vector(D->df(k_vec->get_x()), D->df(k_vec->get_y()), D->df(k_vec->get_z()));

        dDdw = D->df(w);
        dDdkx = D->df(kx); dDdky = D->df(ky); dDdkz = D->df(kz);
        dDdx = D->df(x); dDdy = D->df(y); dDdz = D->df(z);

        if (graph::pseudo_variable_cast(x).get()) {
            dkdx = dkdx->remove_pseudo(); dkdy = dkdy->remove_pseudo(); dkdz = dkdz->remove_pseudo();
            dDdk_vec = dDdk_vec->remove_pseudo();
            dDdw = dDdw->remove_pseudo();
            dDdkx = dDdkx->remove_pseudo(); dDdky = dDdky->remove_pseudo(); dDdkz = dDdkz->remove_pseudo();
            dDdx = dDdx->remove_pseudo(); dDdy = dDdy->remove_pseudo(); dDdz = dDdz->remove_pseudo();
        }

        dxdt = -dDdkx/dDdw;
        dydt = -dDdky/dDdw;
        dzdt = -dDdkz/dDdw;
        dkxdt = (dDdx - dDdk_vec->dot(dkdx))/dDdw;
        dkydt = (dDdy - dDdk_vec->dot(dkdy))/dDdw;
        dkzdt = (dDdz - dDdk_vec->dot(dkdz))/dDdw;
    }
};

// Ray variables in the input order of solver_interface (solver.hpp:304-313).
template<typename T>
struct ray_variables {
    leaf<T> t, w, x, y, z, kx, ky, kz;
    ray_variables() {
        w = graph::variable<T> (1, "\\omega");
        kx = graph::variable<T> (1, "k_{x}"); ky = graph::variable<T> (1, "k_{y}"); kz = graph::variable<T> (1, "k_{z}");
        x = graph::variable<T> (1, "x"); y = graph::variable<T> (1, "y"); z = graph::variable<T> (1, "z");
        t = graph::variable<T> (1, "t");
    }
    std::vector<leaf<T>> inputs() const { return {t, w, x, y, z, kx, ky, kz}; }
};

//  solver::newton (newton.hpp:34-51): setter x - step*func/func->df(x), output func*func.
template<typename T>
work_item<T> make_loss_kernel(const ray_variables<T> &v, leaf<T> func, const int var, const T step) {
    const leaf<T> unknowns[7] = {v.w, v.kx, v.ky, v.kz, v.x, v.y, v.z};     // var: 0 w, 1 kx, 2 ky, 3 kz, 4 x, 5 y, 6 z
    leaf<T> target = unknowns[var];
    return work_item<T> (v.inputs(), {func*func}, {{target - step*func/func->df(target), target}});
}

//  converge_item::run, workflow.hpp:179-205 (max over the shard: cpu_context.hpp:306-322).
template<typename T>
size_t converge(const work_item<T> &item, const size_t n, std::vector<T *> columns, T *residual,
                const T tolerance, const size_t max_iterations, T *last) {
    auto max_kernel = [&] () -> T {
        item.run(n, columns, {residual});
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
// solver::rk4 ctor, solver.hpp:777-870, and the `solver_kernel` item of
// solver_interface::compile, solver.hpp:303-349.
// ---------------------------------------------------------------------------
template<typename T>
struct rk4_step {
    leaf<T> kx_next, ky_next, kz_next, x_next, y_next, z_next, t_next, residual;
};

template<typename T>
rk4_step<T> make_rk4_step(const ray_variables<T> &v, equilibrium_base<T> &eq, leaf<T> dt,
                          dispersion_interface<T> &D) {
    auto kx1 = dt*D.dkxdt, ky1 = dt*D.dkydt, kz1 = dt*D.dkzdt;
    auto x1 = dt*D.dxdt, y1 = dt*D.dydt, z1 = dt*D.dzdt;

    dispersion_interface<T> D2(v.w,
                               graph::pseudo_variable(v.kx + kx1/2.0),
                               graph::pseudo_variable(v.ky + ky1/2.0),
                               graph::pseudo_variable(v.kz + kz1/2.0),
                               graph::pseudo_variable(v.x + x1/2.0),
                               graph::pseudo_variable(v.y + y1/2.0),
                               graph::pseudo_variable(v.z + z1/2.0), eq, D.function);
    auto kx2 = dt*D2.dkxdt, ky2 = dt*D2.dkydt, kz2 = dt*D2.dkzdt;
    auto x2 = dt*D2.dxdt, y2 = dt*D2.dydt, z2 = dt*D2.dzdt;

    dispersion_interface<T> D3(v.w,
                               graph::pseudo_variable(v.kx + kx2/2.0),
                               graph::pseudo_variable(v.ky + ky2/2.0),
                               graph::pseudo_variable(v.kz + kz2/2.0),
                               graph::pseudo_variable(v.x + x2/2.0),
                               graph::pseudo_variable(v.y + y2/2.0),
                               graph::pseudo_variable(v.z + z2/2.0), eq, D.function);
    auto kx3 = dt*D3.dkxdt, ky3 = dt*D3.dkydt, kz3 = dt*D3.dkzdt;
    auto x3 = dt*D3.dxdt, y3 = dt*D3.dydt, z3 = dt*D3.dzdt;

    auto t_next = v.t + dt;

    dispersion_interface<T> D4(v.w,
                               graph::pseudo_variable(v.kx + kx3),
                               graph::pseudo_variable(v.ky + ky3),
                               graph::pseudo_variable(v.kz + kz3),
                               graph::pseudo_variable(v.x + x3),
                               graph::pseudo_variable(v.y + y3),
                               graph::pseudo_variable(v.z + z3), eq, D.function);
    auto kx4 = dt*D4.dkxdt, ky4 = dt*D4.dkydt, kz4 = dt*D4.dkzdt;
    auto x4 = dt*D4.dxdt, y4 = dt*D4.dydt, z4 = dt*D4.dzdt;

    auto kx_next = v.kx + (kx1 + 2.0*(kx2 + kx3) + kx4)/6.0;
    auto ky_next = v.ky + (ky1 + 2.0*(ky2 + ky3) + ky4)/6.0;
    auto kz_next = v.kz + (kz1 + 2.0*(kz2 + kz3) + kz4)/6.0;
    auto x_next = v.x + (x1 + 2.0*(x2 + x3) + x4)/6.0;
    auto y_next = v.y + (y1 + 2.0*(y2 + y3) + y4)/6.0;
    auto z_next = v.z + (z1 + 2.0*(z2 + z3) + z4)/6.0;

    auto residual = D.D*D.D;                                                // dispersion.hpp:1474
    return {kx_next, ky_next, kz_next, x_next, y_next, z_next, t_next, residual};
}

template<typename T>
work_item<T> make_solver_kernel(const ray_variables<T> &v, equilibrium_base<T> &eq, const T dt_value,
                                dispersion_interface<T> &D) {
    const rk4_step<T> s = make_rk4_step<T> (v, eq, graph::constant<T> (dt_value), D);
    return work_item<T> (v.inputs(), {s.residual},
                         {{s.kx_next, v.kx}, {s.ky_next, v.ky}, {s.kz_next, v.kz},
                          {s.x_next, v.x}, {s.y_next, v.y}, {s.z_next, v.z}, {s.t_next, v.t}});
}

// ---------------------------------------------------------------------------
// solver::adaptive_rk4, solver.hpp:877-1006: the RK4 step with dt a VARIABLE, and before every
// step a Newton converge item (newton.hpp:34-51) on the two unknowns (dt, lambda) of
//     loss = 1/dt + lambda*D(next state)^2,
// where the next state enters a second dispersion_interface through pseudo variables
// (solver.hpp:937-947), so that d(loss)/d(dt) sees only the 1/dt term.
// compile() (solver.hpp:951-1003): item 1 = the converge item `loss_kernel` over
// {t,w,x,y,z,kx,ky,kz,dt,lambda}; item 2 = `solver_kernel` over {t,...,kz,dt}.
// ---------------------------------------------------------------------------
template<typename T>
struct adaptive_rk4_items {
    leaf<T> dt, lambda;
    std::unique_ptr<work_item<T>> loss, solver;
};

template<typename T>
adaptive_rk4_items<T> make_adaptive_rk4(const ray_variables<T> &v, equilibrium_base<T> &eq, dispersion_interface<T> &D) {
    adaptive_rk4_items<T> items;
    items.dt = graph::variable<T> (1, "dt");
    items.lambda = graph::variable<T> (1, "\\lambda");
    const rk4_step<T> s = make_rk4_step<T> (v, eq, items.dt, D);
    dispersion_interface<T> next(v.w, graph::pseudo_variable(s.kx_next), graph::pseudo_variable(s.ky_next),
                                 graph::pseudo_variable(s.kz_next), graph::pseudo_variable(s.x_next),
                                 graph::pseudo_variable(s.y_next), graph::pseudo_variable(s.z_next), eq, D.function);
    auto loss = graph::one<T> ()/items.dt + items.lambda*next.D*next.D;
    std::vector<leaf<T>> inputs = v.inputs();
    inputs.push_back(items.dt);
    inputs.push_back(items.lambda);
    const T step = static_cast<T> (1.0);
    items.loss.reset(new work_item<T> (inputs, {loss*loss},
                                       {{items.dt - step*loss/loss->df(items.dt), items.dt},
                                        {items.lambda - step*loss/loss->df(items.lambda), items.lambda}}));
    inputs.pop_back();
    items.solver.reset(new work_item<T> (inputs, {s.residual},
                                         {{s.kx_next, v.kx}, {s.ky_next, v.ky}, {s.kz_next, v.kz},
                                          {s.x_next, v.x}, {s.y_next, v.y}, {s.z_next, v.z}, {s.t_next, v.t}}));
    return items;
}

// ---------------------------------------------------------------------------
// The absorption pass of xrays (graph_driver/xrays.cpp:599-665, :1101): complex base type,
// SAFE_MATH = true.  Constants: dispersion::physics, dispersion.hpp:490-503 (complex arithmetic
// when T is complex: `c` is a complex quotient of a complex square root, as there).
// ---------------------------------------------------------------------------
template<typename T>
struct absorption_constants {
    const T epsilon0 = 8.8541878138E-12;
    const T mu0 = M_PI*4.0E-7;
    const T q = 1.602176634E-19;
    const T me = 9.1093837015E-31;
    const T c = static_cast<T> (1.0)/std::sqrt(epsilon0*mu0);
};

//  dispersion::cold_plasma_expansion::D, dispersion.hpp:1040-1095.
template<typename T, bool S>
leaf<T, S> cold_plasma_expansion_D(leaf<T, S> w, vec3<T, S> k_vec, leaf<T, S> x, leaf<T, S> y, leaf<T, S> z,
                                   equilibrium_base<T, S> &eq) {
    const absorption_constants<T> p;
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto b_len = b_vec->length();
    auto b_hat = b_vec/b_len;
    auto ne = eq.get_electron_density(x, y, z);
    auto te = eq.get_electron_temperature(x, y, z);

//  `ve` is built and dropped there as well (:1054-1056); it leaves no node in the item.
    auto ec = p.q*b_len/(p.me*p.c);                                         // :348-353
    auto wpe2 = ne*p.q*p.q/(p.epsilon0*p.me*p.c*p.c);                       // :326-332

    auto P = wpe2/(w*w);
    auto q = P/(2.0*(1.0 + ec/w));

    auto n = k_vec/w;
    auto n2 = n->dot(n);
    auto npara = n->dot(b_hat);
    auto npara2 = npara*npara;
    auto nperp = b_hat->cross(n);
    auto nperp2 = nperp->dot(nperp);
    auto n2nperp2 = n2*nperp2;

    auto q_func = 1.0 - 2.0*q;
    auto n_func = n2 + npara2;
    auto p_func = 1.0 - P;

    auto gamma1 = (1.0 - q)*n2nperp2
                + p_func*(n2*npara2 - (1.0 - q)*n_func)
                + q_func*(p_func - nperp2);
    auto gamma0 = nperp2*(n2 - 2.0*q_func) + p_func*(2.0*q_func - n_func);

    return -P/2.0*(1.0 + ec/w)*gamma0 + (1.0 - ec*ec/(w*w))*gamma1;
}

template<typename T, bool S> std::vector<leaf<T, S>> *absorption_debug = nullptr;

//  dispersion::z_erfi::Z, dispersion.hpp:289-297.
template<typename T, bool S>
leaf<T, S> z_erfi_Z(leaf<T, S> zeta) {
    return -std::sqrt(M_PI)*graph::exp(-zeta*zeta)*(graph::erfi(zeta) - graph::i<T>);
}

//  dispersion::hot_plasma_expansion<T, z_erfi, SAFE_MATH>::D, dispersion.hpp:1230-1300.
template<typename T, bool S>
leaf<T, S> hot_plasma_expansion_D(leaf<T, S> w, vec3<T, S> k_vec, leaf<T, S> x, leaf<T, S> y, leaf<T, S> z,
                                  equilibrium_base<T, S> &eq) {
    const absorption_constants<T> p;
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto b_hat = b_vec->unit();
    auto b_len = b_vec->length();
    auto ne = eq.get_electron_density(x, y, z);
    auto te = eq.get_electron_temperature(x, y, z);

    auto ve = graph::sqrt(static_cast<T> (2.0)*p.q*te/p.me);

    auto ec = p.q*b_len/(p.me*p.c);
    auto wpe2 = ne*p.q*p.q/(p.epsilon0*p.me*p.c*p.c);

    auto P = wpe2/(w*w);
    auto q = P/(2.0*(1.0 + ec/w));

    auto n = k_vec/w;
    auto n2 = n->dot(n);
    auto npara = b_hat->dot(n);
    auto npara2 = npara*npara;
    auto nperp = b_hat->cross(n);
    auto nperp2 = nperp->dot(nperp);

    auto vtnorm = ve/p.c;

    auto zeta = (1.0 - ec/w)/(npara*vtnorm);
    auto Z_func = z_erfi_Z<T, S> (zeta);
    if (absorption_debug<T, S>) *absorption_debug<T, S> = {zeta, Z_func, ec, vtnorm, npara, P, te, ne};

    auto q_func = 1.0 - 2.0*q;
    auto n_func = n2 + npara2;
    auto n2nperp2 = n2*nperp2;
    auto p_func = 1.0 - P;

    auto gamma5 = P*(n2*npara2 - (1.0 - q)*n_func + q_func);
    auto gamma2 = P*w/ec*nperp2*(n2 - q_func)
                + P*P*w*w/(4.0*ec*ec)*(n_func - 2.0*q_func)*nperp2/npara2;
    auto gamma1 = (1.0 - q)*n2nperp2
                + p_func*(n2*npara2 - (1.0 - q)*n_func)
                + q_func*(p_func - nperp2);

    return -(1.0 + ec/w)*npara*vtnorm *
           (gamma1 + gamma2 + nperp2/(2.0*npara)*(w*w/(ec*ec))*vtnorm*zeta*gamma5)*(1.0/Z_func + zeta);
}

//  dispersion::hot_plasma<T, z_erfi, SAFE_MATH>::D, dispersion.hpp:1106-1163.
template<typename T, bool S>
leaf<T, S> hot_plasma_D(leaf<T, S> w, vec3<T, S> k_vec, leaf<T, S> x, leaf<T, S> y, leaf<T, S> z,
                        equilibrium_base<T, S> &eq) {
    const absorption_constants<T> p;
    auto b_vec = eq.get_magnetic_field(x, y, z);
    auto b_len = b_vec->length();
    auto b_hat = b_vec/b_len;
    auto ne = eq.get_electron_density(x, y, z);
    auto te = eq.get_electron_temperature(x, y, z);

    auto ve = graph::sqrt(static_cast<T> (2.0)*p.q*te/p.me)/p.c;

    auto ec = p.q*b_len/(p.me*p.c);
    auto wpe2 = ne*p.q*p.q/(p.epsilon0*p.me*p.c*p.c);

    auto P = wpe2/(w*w);
    auto q = P/(2.0*(1.0 + ec/w));

    auto n = k_vec/w;
    auto n2 = n->dot(n);
    auto npara = n->dot(b_hat);
    auto npara2 = npara*npara;
    auto nperp = b_hat->cross(n);
    auto nperp2 = nperp->dot(nperp);

    auto zeta = (1.0 - ec/w)/(npara*ve);
    auto Z_func = z_erfi_Z<T, S> (zeta);
    auto zeta_func = 1.0 + zeta*Z_func;
    auto F = ve*zeta*w/(2.0*npara*ec);
    auto isigma = P*Z_func/(2.0*npara*ve);

    auto q_func = 1.0 - 2.0*q;
    auto n_func = n2 + npara2;
    auto p_func = 1.0 - P;

    auto gamma5 = n2*npara2 - (1.0 - q)*n_func + q_func;
    auto gamma2 = (n2 - q_func)
                + P*w/(4.0*ec*npara2)*(n_func - 2.0*q_func);
    auto gamma1 = nperp2*((1.0 - q)*n2 - q_func)
                + p_func*(n2*npara2 - (1.0 - q)*n_func + q_func);
    auto gamma0 = nperp2*(n2 - 2.0*q_func) + p_func*(2.0*q_func - n_func);

    return isigma*gamma0 + gamma1 + nperp2*P*w/ec*zeta_func*(gamma2 + gamma5*F);
}

//  absorption::root_finder ctor, absorption.hpp:146-226: three items in the order manager::run executes
//  them — `root_find_init_kernel` (kamp <- 0), the converge item `loss_kernel` of solver::newton
//  (newton.hpp:34-51) on the hot-plasma D(k + kamp k_hat) with kamp the unknown, `final_kamp`
//  (kamp <- |k| + kamp).
template<typename T, bool S>
struct root_finder_items {
    leaf<T, S> kamp, w, kx, ky, kz, x, y, z, t;
    std::unique_ptr<work_item<T, S>> init, loss, final_kamp;

    leaf<T, S> D, klen;

    explicit root_finder_items(equilibrium_base<T, S> &eq, const size_t size = 1) {
        w = graph::variable<T, S> (size, "\\omega");
        kx = graph::variable<T, S> (size, "k_{x}");
        ky = graph::variable<T, S> (size, "k_{y}");
        kz = graph::variable<T, S> (size, "k_{z}");
        x = graph::variable<T, S> (size, "x");
        y = graph::variable<T, S> (size, "y");
        z = graph::variable<T, S> (size, "z");
        t = graph::variable<T, S> (size, "t");
        kamp = graph::variable<T, S> (size, "kamp");

        auto kvec = kx*eq.esup(0) + ky*eq.esup(1) + kz*eq.esup(2);
        klen = kvec->length();
        auto kamp_vec = kamp*kvec/klen;

        std::vector<leaf<T, S>> inputs = {kamp, kx, ky, kz, x, y, z};
        init.reset(new work_item<T, S> (inputs, {}, {{graph::zero<T, S> (), kamp}}));

        std::vector<leaf<T, S>> newton_inputs = inputs;
        newton_inputs.push_back(t);
        newton_inputs.push_back(w);
        D = hot_plasma_D<T, S> (w, kvec + kamp_vec, x, y, z, eq);
        const T step = 1.0;
        loss.reset(new work_item<T, S> (newton_inputs, {D*D}, {{kamp - step*D/D->df(kamp), kamp}}));

        final_kamp.reset(new work_item<T, S> (inputs, {}, {{klen + kamp, kamp}}));
    }
};

//  absorption::weak_damping ctor, absorption.hpp:346-432: the one work item of the pass,
//  `weak_damping_kimg_kernel`, inputs {kamp, kx, ky, kz, x, y, z, t, w}, no outputs, one setter
//      kamp <- |k| - Dw/(k_hat . grad_k Dc).
template<typename T, bool S>
struct weak_damping_item {
    leaf<T, S> kamp, w, kx, ky, kz, x, y, z, t;
    leaf<T, S> kamp1;
    graph::input_nodes<T, S> inputs;
    graph::map_nodes<T, S> setters;

    explicit weak_damping_item(equilibrium_base<T, S> &eq, const size_t size = 1) {
        w = graph::variable<T, S> (size, "\\omega");                         // xrays.cpp:622-630
        kx = graph::variable<T, S> (size, "k_{x}");
        ky = graph::variable<T, S> (size, "k_{y}");
        kz = graph::variable<T, S> (size, "k_{z}");
        x = graph::variable<T, S> (size, "x");
        y = graph::variable<T, S> (size, "y");
        z = graph::variable<T, S> (size, "z");
        t = graph::variable<T, S> (size, "t");
        kamp = graph::variable<T, S> (size, "kamp");

        auto k_vec = kx*eq.esup(0) + ky*eq.esup(1) + kz*eq.esup(2);
        auto k_unit = k_vec->unit();

        auto Dc = cold_plasma_expansion_D<T, S> (w, k_vec, x, y, z, eq);
        auto Dw = hot_plasma_expansion_D<T, S> (w, k_vec, x, y, z, eq);

        kamp1 = k_vec->length()
              - Dw/k_unit->dot(Dc->df(kx)*eq.esup(0) +
                               Dc->df(ky)*eq.esup(1) +
                               Dc->df(kz)*eq.esup(2));

        inputs = {graph::variable_cast(kamp), graph::variable_cast(kx), graph::variable_cast(ky),
                  graph::variable_cast(kz), graph::variable_cast(x), graph::variable_cast(y),
                  graph::variable_cast(z), graph::variable_cast(t), graph::variable_cast(w)};
        setters = {{kamp1, graph::variable_cast(kamp)}};
    }
};

//  bin_power, graph_driver/xrays.cpp:674-790: the `power` item (real base type, SAFE_MATH off).
//  efit keeps generic::get_x/y/z (equilibrium.hpp:431-470): the coordinates themselves.
template<typename T>
struct power_item {
    leaf<T> x, y, z, x_last, y_last, z_last, kamp, power, k_sum, d_power;
    graph::input_nodes<T> inputs;
    graph::output_nodes<T> outputs;
    graph::map_nodes<T> setters;

    explicit power_item(const size_t size = 1) {
        x = graph::variable<T> (size, "x");
        y = graph::variable<T> (size, "y");
        z = graph::variable<T> (size, "z");
        x_last = graph::variable<T> (size, "x_last");
        y_last = graph::variable<T> (size, "y_last");
        z_last = graph::variable<T> (size, "z_last");
        kamp = graph::variable<T> (size, "kamp");
        power = graph::variable<T> (size, static_cast<T> (1.0), "power");
        k_sum = graph::variable<T> (size, static_cast<T> (0.0), "k_sum");

        auto dlvec = graph::vector(x - x_last, y - y_last, z - z_last);
        auto dl = dlvec->length();
        auto kdl = kamp*dl;
        auto k_next = kdl + k_sum;
        auto p_next = graph::exp(-2.0*k_sum);
        d_power = p_next - power;
        d_power = graph::sqrt(d_power*d_power);

        inputs = {graph::variable_cast(x), graph::variable_cast(y), graph::variable_cast(z),
                  graph::variable_cast(x_last), graph::variable_cast(y_last), graph::variable_cast(z_last),
                  graph::variable_cast(kamp), graph::variable_cast(power), graph::variable_cast(k_sum)};
        outputs = {d_power};
        setters = {{x, graph::variable_cast(x_last)}, {y, graph::variable_cast(y_last)},
                   {z, graph::variable_cast(z_last)}, {p_next, graph::variable_cast(power)},
                   {k_next, graph::variable_cast(k_sum)}};
    }
};

#endif /* ref_builders_hpp */

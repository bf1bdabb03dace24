//------------------------------------------------------------------------------
///  @file codegen.hpp
///  @brief Lower a GFIR work item to one CDNA4 (gfx950) HIP kernel.
///
///  The reference emits one line of C++ per DAG node between a backend-written
///  prefix and postfix (jit.hpp:118-194; cuda_context.hpp:713-946).  This
///  lowering keeps the node-for-node arithmetic (one IEEE operation per node,
///  fma only where the graph has an fma_node, no contraction, IEEE divide and
///  sqrt) because results must match gpu::cpu_context, and spends its freedom
///  on the memory side, which is where the MI355X differs from the reference's
///  targets:
///
///  * one wavefront lane owns one ray/particle; every state array is read once
///    into registers (coalesced 8 B/lane SoA loads) and every setter target is
///    written once (cuda_context.hpp:862-946 does the same per thread);
///  * the reference emits each coefficient table as its own array and repeats
///    the full clamp+truncate index expression at every gather
///    (piecewise.hpp:268-303, :1072-1208, no USE_INDEX_CACHE).  Here gathers
///    are grouped by (argument, scale, offset, shape): one index per group —
///    8 per RK4 step instead of 360 — and all tables of one shape are packed
///    AoS `[cell][table]`, so the 45 coefficients a lane needs at its cell sit
///    in consecutive bytes;
///  * packs that fit the LDS budget (the 1-D profile tables) are staged into LDS
///    once per workgroup; the 2-D psi pack stays L2 resident (4 MiB per XCD);
///  * `steps` passes can run inside one launch with the state kept in
///    registers (the reference launches once per step, solver.hpp:382);
///  * small items with an output get two more entry points: `<name>_max` folds the
///    max of the last output into the launch (wave shuffle + one atomic per
///    workgroup) — what create_max_call needs, where cuda_context.hpp:540-576
///    runs a second kernel over the output — and `<name>_converge` runs the stall
///    loop of workflow.hpp:179-205 per ray inside the launch.
///
///  Pieces: options.hpp (knobs, cache hash), schedule.hpp (emission order),
///  tables.hpp (compaction, packs, LDS staging), parking.hpp (values that wait
///  in LDS), prelude.hpp (device helpers: division, window checks, pow),
///  segments.hpp (items cut into several kernels, the redo launch), asm_body.hpp
///  (the pass of a large fp64 item as gfx950 assembly with a register assignment
///  of its own: the RK4 step's default), and lower() below, which writes the
///  kernel text.
//------------------------------------------------------------------------------
#ifndef gfhip_codegen_hpp
#define gfhip_codegen_hpp

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "asm_body.hpp"
#include "gfir_item.hpp"
#include "options.hpp"
#include "parking.hpp"
#include "prelude.hpp"
#include "schedule.hpp"
#include "tables.hpp"

namespace gfhip {

struct lowered {
    std::string source;
    std::string kernel_name;
    std::vector<pack> packs;
    std::vector<bool> input_written;    ///< input i is the target of a setter
    std::vector<int> table_parent;      ///< -1 = stored; else the table it is an exact multiple of
    std::vector<double> table_factor;
    uint32_t block_size = 256;
    size_t lds_bytes = 0;
    uint32_t park_slots = 0;            ///< LDS slots used for parked values
    bool assembly = false;              ///< the pass is the assembly statement of asm_body.hpp
    bool has_converge = false;          ///< the module also holds `<name>_converge`
    bool has_max = false;               ///< the module also holds `<name>_max`
    uint32_t batch = 0;                 ///< the module also holds `<name>_batch`: up to this many passes per launch, each with its own max
    uint64_t hash = 0;
};

///  Entry points of one item's module.
enum class entry { plain, max, converge, batch };

///  What a lowered item is within a segmented item whose out-of-window lanes are redone by a separate
///  launch instead of an IEEE function compiled into every kernel (segments.hpp):
///    middle  a segment that is not the last: a lane that fails a check raises its per-ray flag;
///    last    the last segment: a lane that failed a check here or in an earlier segment skips its stores
///            and appends its ray index to the redo list (one atomicAdd per wavefront that has such lanes);
///    redo    the WHOLE item with the compiler's division over the rays of the redo list, from the state
///            the segments left untouched.
enum class piece_role { none, middle, last, redo };

struct piece_info {
    piece_role role = piece_role::none;
    bool scheduled = false;                     ///< the piece arrives in emission order (gf_hip.cpp cut it from the scheduled item)
    std::vector<bool> output_handed_over;       ///< per output: a hand-over value (no stored-value checks apply to it)
    std::vector<bool> symbol_after_division;    ///< per symbol: a handed-over value that depends on a quotient of an earlier segment
                                                ///< (a zero of it stored here may carry the wrong sign, like a zero computed here)
};

//------------------------------------------------------------------------------
///  @brief Writes the kernel text of one lowered item.
//------------------------------------------------------------------------------
struct kernel_writer {
    std::ostringstream &s;
    const item &it;
    const codegen_options &opt;
    const lowered &out;
    const std::vector<int> &parent;                 ///< table compaction (tables.hpp)
    const std::vector<double> &factor;
    const std::vector<uint32_t> &table_pack, &table_column;
    const std::vector<park_plan> &plan;             ///< LDS parking (parking.hpp)
    const size_t lds_used;                          ///< staged packs + parking slots, bytes
    const size_t park_offset;
    const uint32_t park_slots;
    const bool use_shared;                          ///< shared-reciprocal division with the IEEE second body
    const std::vector<bool> &after_division;        ///< node depends on the result of a division
    const piece_info &piece;                        ///< role within a segmented item with a redo launch
    const std::string &asm_statement;               ///< not empty: the shared-reciprocal body as one assembly statement (asm_body.hpp)

    const bool f64 = it.base_is_f64();
    const bool cx = it.is_complex();                ///< values are gf_complex (prelude.hpp)
    const bool safe = it.safe_math();               ///< SAFE_MATH guards
    const bool generic = cx || safe || it.has_random();     ///< operations go through the gf_* names
    const bool track_numerators = opt.division == division_mode::checked && it.base_is_f64();   // fp32 quotients go through fp64: nothing to track
    const bool fixup = division_fixup(it, opt);
    const bool fast = opt.division == division_mode::fast;      ///< tolerance mode: no residual step, no checks
    const char *real = f64 ? "double" : "float";
    const std::string sfx = f64 ? "" : "f";
    const size_t esize = it.element_size();
    const size_t node_count = it.code.size();

    std::string literal(const double v) const {
        char buf[64];
        if (f64) {
            std::snprintf(buf, sizeof(buf), "%a", v);
        } else {
            std::snprintf(buf, sizeof(buf), "%af", static_cast<double> (static_cast<float> (v)));
        }
        std::string s(buf);
        if (s.find("inf") != std::string::npos || s.find("nan") != std::string::npos) {
            s = std::string("((") + real + ")" + (v != v ? "__builtin_nan(\"\")" : (v > 0 ? "__builtin_inf()" : "-__builtin_inf()")) + ")";
        }
        return s;
    }

//  A constant of the item's value type.
    std::string value_literal(const double re, const double im = 0.0) const {
        return cx ? "real(" + literal(re) + ", " + literal(im) + ")" : literal(re);
    }
    std::string call(const char *builtin, const char *generic_name) const {
        return generic ? std::string(generic_name) : std::string(builtin) + sfx;
    }

//  The node-for-node body.  `shared` = divisions through a reciprocal shared by all
//  divisions with the same denominator (gf_rcp/gf_div in the prelude) plus the checks that
//  tell whether the lane may keep that result; otherwise the compiler's IEEE division.
    void body(const bool shared) {
        typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, double, double, double, double> group_key;
        std::map<group_key, std::string> groups;
        std::map<uint32_t, bool> reciprocal_done;
        std::set<std::string> coefficients;
        size_t group_count = 0;
//  Current name of every value (changes when a parked value is reloaded).
        std::vector<std::string> name(node_count);
        for (size_t v = 0; v < node_count; v++) name[v] = "r" + std::to_string(v);
        auto N = [&] (const uint32_t v) -> const std::string & { return name[v]; };
//  Index of one dimension of a gather: clamp((x - offset)/scale) truncated (compile_index,
//  piecewise.hpp:26-65).  The shared form divides through the correctly rounded literal
//  1/scale (gf_div's residual step makes the quotient the correctly rounded one) and hands the
//  quotient to the finite check: an overflowing or non-finite argument comes out of the shared
//  sequence as a NaN, which clamps to cell 0 where IEEE division's infinity clamps to the last.
        size_t quotient_count = 0;
        auto index_expression = [&] (const uint32_t arg, const double scale, const double offset,
                                     const uint32_t length) -> std::string {
            std::ostringstream e;
            e << "static_cast<unsigned int> (__builtin_fmin" << sfx << "(__builtin_fmax" << sfx << "(";
            if (shared) {
                const std::string quotient = "x" + std::to_string(quotient_count++);
                if (f64) {
                    s << "                const real " << quotient << " = gf_div(" << N(arg) << " - " << literal(offset) << ", " << literal(scale) << ", "
                      << literal(1.0/scale) << ");\n";
                    if (!fast) s << "                vmax = __builtin_elementwise_maximum(vmax, gf_magnitude(" << quotient << "));\n";
                } else {
//  fp32: the reciprocal is the double nearest to 1/scale (prelude.hpp: quotients through fp64)
                    char wide[64];
                    std::snprintf(wide, sizeof(wide), "%a", 1.0/static_cast<double> (static_cast<float> (scale)));
                    s << "                const real " << quotient << " = gf_div(" << N(arg) << " - " << literal(offset) << ", " << literal(scale) << ", "
                      << wide << ");\n";
                }
                e << quotient;
            } else {
                const std::string quotient = "(" + N(arg) + " - " + value_literal(offset) + ")/" + value_literal(scale);
                e << (cx ? "gf_real(" + quotient + ")" : quotient);     // compile_index takes real(...) of a complex quotient
            }
            e << ", " << literal(0.0) << "), " << literal(static_cast<double> (length - 1)) << "))";
            return e.str();
        };

        const char *ind = "                ";
        std::map<size_t, std::vector<uint32_t>> reloads_at_position;
        for (size_t v = 0; v < node_count; v++) {
            if (!plan[v].parked) continue;
            for (auto &kv : plan[v].reload_at) reloads_at_position[kv.first].push_back(static_cast<uint32_t> (v));
        }
        auto reload = [&] (const size_t position) {
            auto found = reloads_at_position.find(position);
            if (found == reloads_at_position.end()) return;
            for (const uint32_t v : found->second) {
                name[v] = "r" + std::to_string(v) + "p" + std::to_string(plan[v].reload_at.at(position));
                s << ind << "const real " << name[v] << " = park_read[" << plan[v].slot*out.block_size << "u];\n";
            }
        };
//  Value of table `t` at the cell of index group `group`: a load for stored tables, an exact
//  multiple of the parent's value otherwise; one definition per (group, table).
        std::function<std::string(const std::string &, uint32_t)> value_at =
            [&] (const std::string &group, const uint32_t t) -> std::string {
            const std::string value_name = "c" + group.substr(1) + "_" + std::to_string(t);
            if (coefficients.insert(value_name).second) {
                if (parent[t] >= 0) {
                    const std::string from = value_at(group, static_cast<uint32_t> (parent[t]));
                    s << ind << "const real " << value_name << " = " << literal(factor[t]) << "*" << from << ";\n";
                } else {
//  `group` points at the cell: the column is an immediate offset of the load (one address per group, not per table).
                    s << ind << "const real " << value_name << " = " << group << "[" << table_column[t] << "u];\n";
                }
            }
            return value_name;
        };
        std::map<uint32_t, std::pair<std::string, uint32_t>> deferred;     // gather node -> (group, table)
        auto define = [&] (const uint32_t v) {
            auto found = deferred.find(v);
            if (found == deferred.end()) return;
            const std::string value = value_at(found->second.first, found->second.second);
            s << ind << "const real r" << v << " = " << value << ";\n";
            deferred.erase(found);
        };
        for (size_t i = 0; i < it.code.size(); i++) {
            const gfir_instruction &c = it.code[i];
            reload(i);
            {
                const uint32_t operands[3] = {c.a, c.b, c.c};
                for (int k = 0; k < operand_count(c.op); k++) define(operands[k]);
            }
            switch (c.op) {
                case GFIR_CONST:
                    s << ind << "const real r" << i << " = " << value_literal(c.imm[0], c.imm[1]) << ";\n";
                    break;
                case GFIR_RANDOM:
                    s << ind << "const real r" << i << " = gf_from_base(static_cast<base> (gf_random(random_state)));\n";
                    break;
                case GFIR_INPUT:
                    s << ind << "const real r" << i << " = v" << c.a << ";\n";
                    break;
                case GFIR_ADD:
                    s << ind << "const real r" << i << " = " << N(c.a) << " + " << N(c.b) << ";\n";
                    break;
                case GFIR_SUB:
                    s << ind << "const real r" << i << " = " << N(c.a) << " - " << N(c.b) << ";\n";
                    break;
                case GFIR_MUL:
//  SAFE_MATH: (l == 0 || r == 0) ? 0 : l*r, arithmetic.hpp:2534-2557
                    s << ind << "const real r" << i << " = ";
                    if (safe) s << "(" << N(c.a) << " == " << value_literal(0.0) << " || " << N(c.b) << " == " << value_literal(0.0) << ") ? " << value_literal(0.0) << " : ";
                    s << N(c.a) << "*" << N(c.b) << ";\n";
                    break;
                case GFIR_DIV:
                    if (shared) {
                        if (!reciprocal_done[c.b]) {
                            reciprocal_done[c.b] = true;
                            s << ind << (f64 ? "const real q" : "const double q") << c.b << " = gf_rcp(" << N(c.b) << ");\n";
                            if (!fast) {
                                s << ind << "dmax = __builtin_elementwise_maximum(dmax, gf_magnitude(" << N(c.b) << "));\n";
                                s << ind << "dmin = __builtin_elementwise_minimum(dmin, gf_magnitude(" << N(c.b) << "));\n";
                            }
                        }
                        if (track_numerators) {
                            s << ind << "nmin = __builtin_elementwise_min(nmin, gf_numerator_key(" << N(c.a) << "));\n";
                        }
                        s << ind << "const real r" << i << " = gf_div(" << N(c.a) << ", " << N(c.b) << ", q" << c.b << ");\n";
                    } else {
//  SAFE_MATH: l == 0 ? 0 : l/r, arithmetic.hpp:3526-3541
                        s << ind << "const real r" << i << " = ";
                        if (safe) s << N(c.a) << " == " << value_literal(0.0) << " ? " << value_literal(0.0) << " : ";
                        s << N(c.a) << "/" << N(c.b) << ";\n";
                    }
                    break;
                case GFIR_FMA:
//  SAFE_MATH: (l == 0 || m == 0) ? r : fma(l, m, r), arithmetic.hpp:5101-5127
                    s << ind << "const real r" << i << " = ";
                    if (safe) s << "(" << N(c.a) << " == " << value_literal(0.0) << " || " << N(c.b) << " == " << value_literal(0.0) << ") ? " << N(c.c) << " : ";
                    s << call("__builtin_fma", "gf_fma") << "(" << N(c.a) << ", " << N(c.b) << ", " << N(c.c) << ");\n";
                    break;
                case GFIR_SQRT:
//  Inside the checked window the compiler's 18-instruction sqrt reduces to its 9-instruction core (prelude.hpp,
//  gf_sqrt_window): the argument joins the denominators' window check, a lane outside it takes the IEEE pass.
                    if (shared && f64 && !generic && !fast && opt.window_sqrt) {
                        s << ind << "dmax = __builtin_elementwise_maximum(dmax, gf_magnitude(" << N(c.a) << "));\n";
                        s << ind << "dmin = __builtin_elementwise_minimum(dmin, gf_magnitude(" << N(c.a) << "));\n";
                        s << ind << "const real r" << i << " = gf_sqrt_window(" << N(c.a) << ");\n";
                        break;
                    }
//  fp32: the same for sqrtf (16 -> 9 instructions), with extremes of its own: the fp32 window of the denominators
//  is every finite non-zero float, the compiler's sqrtf re-scales below 2^-96.
                    if (shared && !f64 && !generic && !fast && opt.window_sqrt_f32) {
                        s << ind << "smax = __builtin_elementwise_maximum(smax, " << N(c.a) << ");\n";
                        s << ind << "smin = __builtin_elementwise_minimum(smin, " << N(c.a) << ");\n";
                        s << ind << "const real r" << i << " = gf_sqrt_window(" << N(c.a) << ");\n";
                        break;
                    }
                    s << ind << "const real r" << i << " = " << call("__builtin_sqrt", "gf_sqrt") << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_POWI: {
                    s << ind << "const real r" << i << " = " << N(c.a);
                    for (uint32_t k = 1; k < c.aux; k++) s << "*" << N(c.a);
                    s << ";\n";
                    break;
                }
                case GFIR_POW:
//  pow tells -0 from +0 when the exponent is an odd integer (pow(-0, -1) = -inf, pow(+0, -1) = +inf): a base
//  that comes from a shared-reciprocal quotient (whose zero may carry the wrong sign without v_div_fixup) joins
//  the zero check unless the exponent is a constant that is not an odd integer.
                    if (shared && f64 && !fixup && !fast && after_division[c.a]) {
                        const gfir_instruction &e = it.code[c.b];
                        const bool harmless = e.op == GFIR_CONST &&
                                              !(e.imm[0] == std::floor(e.imm[0]) && std::fmod(std::fabs(e.imm[0]), 2.0) == 1.0);
                        if (!harmless) {
                            s << ind << "zmin = __builtin_elementwise_minimum(zmin, gf_magnitude(" << N(c.a) << "));\n";
                        }
                    }
                    if (f64 && !generic && opt.pow_three_halves && it.code[c.b].op == GFIR_CONST && it.code[c.b].imm[0] == 1.5) {
                        if (shared && !fast && opt.window_sqrt) {
//  pow(x, 1.5) of a negative x is NaN either way; of a zero or an infinity (outside the window) the IEEE pass decides.
                            s << ind << "dmax = __builtin_elementwise_maximum(dmax, gf_magnitude(" << N(c.a) << "));\n";
                            s << ind << "dmin = __builtin_elementwise_minimum(dmin, gf_magnitude(" << N(c.a) << "));\n";
                            s << ind << "const real r" << i << " = gf_pow_three_halves_window(" << N(c.a) << ");\n";
                            break;
                        }
                        s << ind << "const real r" << i << " = gf_pow_three_halves(" << N(c.a) << ");\n";
                    } else {
                        s << ind << "const real r" << i << " = " << call("pow", "gf_pow") << "(" << N(c.a) << ", " << N(c.b) << ");\n";
                    }
                    break;
                case GFIR_SIN:
                    s << ind << "const real r" << i << " = " << call("sin", "gf_sin") << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_COS:
                    s << ind << "const real r" << i << " = " << call("cos", "gf_cos") << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_ATAN2:
                    s << ind << "const real r" << i << " = " << call("atan2", "gf_atan2") << "(" << N(c.b) << ", " << N(c.a) << ");\n";
                    break;
                case GFIR_EXP:
//  SAFE_MATH: real(x) < 709.8 ? exp(x) : the base type's largest value, math.hpp:450-471 (the
//  reference prints that bound through an int-valued max_base(), an out-of-range conversion whose
//  result is undefined; the intended value is used here)
                    s << ind << "const real r" << i << " = ";
                    if (safe) s << "gf_real(" << N(c.a) << ") < static_cast<base> (709.8) ? ";
                    s << call("exp", "gf_exp") << "(" << N(c.a) << ")";
                    if (safe) s << " : gf_from_base(" << (f64 ? "__DBL_MAX__" : "__FLT_MAX__") << ")";
                    s << ";\n";
                    break;
                case GFIR_LOG:
                    s << ind << "const real r" << i << " = " << call("log", "gf_log") << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_ERFI:
                    s << ind << "const real r" << i << " = gf_erfi(" << N(c.a) << ");\n";
                    break;
                case GFIR_GATHER1:
                case GFIR_GATHER2: {
                    const table &t = it.tables[c.aux];
                    const bool two = c.op == GFIR_GATHER2;
                    const group_key key(c.a, two ? c.b : GFIR_NONE, t.rows, t.cols,
                                        c.imm[0], c.imm[1], two ? c.imm[2] : 0.0, two ? c.imm[3] : 0.0);
                    auto g = groups.find(key);
                    if (g == groups.end()) {
                        const std::string group_name = "g" + std::to_string(group_count++);
                        const pack &p = out.packs[table_pack[c.aux]];
                        const std::string first = index_expression(c.a, c.imm[0], c.imm[1], two ? t.rows : t.cols);
                        const std::string second = two ? index_expression(c.b, c.imm[2], c.imm[3], t.cols) : std::string();
//  A pointer to the group's cell in its pack ([cell][table], tables.hpp): computed once, every gather of the group
//  is then a load at an immediate offset from it (as an index added to the pack's base each table needed a 32-bit add
//  and a 64-bit address of its own: 61 address instructions for the 27 loads of the fp32 xkorc push).
                        const uint32_t pi = table_pack[c.aux];
                        const std::string base = (out.packs[pi].in_lds ? "lds" : "pack") + std::to_string(pi);
                        s << ind << "const real *const " << group_name << " = " << base << " + static_cast<size_t> ((" << first;
                        if (two) s << "*" << t.cols << "u + " << second;
                        s << ")*" << p.stride << "u);\n";
                        g = groups.insert({key, group_name}).first;
                    }
//  A derived table's value (k*parent) is defined at its first use, not here: next to the
//  parent's load it would make the pass wait for that load at once.  The load stays here.
                    if (parent[c.aux] >= 0 && !plan[i].parked) {
                        uint32_t root = c.aux;
                        while (parent[root] >= 0) root = static_cast<uint32_t> (parent[root]);
                        (void)value_at(g->second, root);
                        deferred[static_cast<uint32_t> (i)] = {g->second, c.aux};
                        break;
                    }
                    const std::string value = value_at(g->second, c.aux);
                    s << ind << "const real r" << i << " = " << value << ";\n";
                    break;
                }
                case GFIR_INDEX1:
                case GFIR_INDEX2: {
//  index_1D/2D_node: an element of an input variable's own buffer, picked by the clamped index
//  (piecewise.hpp:1530-1575, :1899-1990); a plain global load, no packing.
                    const bool two = c.op == GFIR_INDEX2;
                    const std::string first = index_expression(c.a, c.imm[0], c.imm[1], two ? c.reserved : c.aux);
                    const std::string second = two ? index_expression(c.b, c.imm[2], c.imm[3], c.aux) : std::string();
                    s << ind << "const real r" << i << " = in" << c.c << "[" << first;
                    if (two) s << "*" << c.aux << "u + " << second;
                    s << "];\n";
                    break;
                }
                default:
                    s << ind << "#error unsupported GFIR op\n";
            }
            if (plan[i].parked) {
                s << ind << "park[" << plan[i].slot*out.block_size << "u] = r" << i << ";\n";
            }
        }
        reload(node_count);
        for (auto &st : it.setters) define(st.value);
        for (auto o : it.outputs) define(o);
        for (size_t k = 0; k < it.setters.size(); k++) {
            s << ind << "sv" << k << " = " << N(it.setters[k].value) << ";\n";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << ind << "so" << o << " = " << N(it.outputs[o]) << ";\n";
        }
    }

//  The pass with the compiler's IEEE division, as a function of its own: a lane that failed a
//  check of the shared-reciprocal body calls it.  Not inlined, so that the kernel's hot path
//  keeps the register allocation and schedule it has without it (inlined as a second body it
//  cost the RK4 kernel 4-8 %: spills and 28 B of scratch per lane on the hot path).
    void ieee_function() {
        const std::string results = out.kernel_name + "_results";
        s << "struct " << results << " {";
        for (size_t k = 0; k < it.setters.size(); k++) s << " real sv" << k << ";";
        for (size_t o = 0; o < it.outputs.size(); o++) s << " real so" << o << ";";
        s << " };\n";
        s << "static __device__ __attribute__((noinline, cold)) " << results << "\n" << out.kernel_name << "_ieee(";
        for (size_t i = 0; i < it.symbols.size(); i++) s << "const real v" << i << ", ";
        for (size_t p = 0; p < out.packs.size(); p++) {
            s << "const real *__restrict__ " << (out.packs[p].in_lds ? "lds" : "pack") << p << ", ";
        }
        for (size_t i = 0; i < it.symbols.size(); i++) {               // buffers that index nodes read
            if (it.indexed_length(static_cast<uint32_t> (i))) s << "const real *in" << i << ", ";
        }
        if (park_slots) s << "park_t *park, park_t *park_read, ";
        s << "const int) {\n";
        s << "            " << results << " redo;\n";
        for (size_t k = 0; k < it.setters.size(); k++) s << "            real sv" << k << ";\n";
        for (size_t o = 0; o < it.outputs.size(); o++) s << "            real so" << o << ";\n";
        s << "            {\n";
        body(false);
        s << "            }\n";
        for (size_t k = 0; k < it.setters.size(); k++) s << "            redo.sv" << k << " = sv" << k << ";\n";
        for (size_t o = 0; o < it.outputs.size(); o++) s << "            redo.so" << o << " = so" << o << ";\n";
        s << "            return redo;\n}\n\n";
    }

    void kernel(const entry which) {
        signature(which);
        lds_setup();
        tile_open(which);
        if (which == entry::batch) {
//  `<name>_batch`: up to opt.converge_batch passes per launch on state kept in registers, EVERY pass with
//  its own max (converge loops decide pass by pass, gf_hip.cpp), the state of the beginning of the launch
//  saved in the undo arrays first (the loop may turn out to have ended inside the batch).
            for (uint32_t b = 0; b < opt.converge_batch; b++) {
                batch_pass = static_cast<int> (b);
                pass_open(which);
                pass(which);
            }
            batch_pass = -1;
        } else {
            pass_open(which);
            pass(which);
        }
        stores(which);
    }
    int batch_pass = -1;                            ///< which unrolled pass of `<name>_batch` is being written

//  Kernel name, arguments.
    void signature(const entry which) {
        s << "extern \"C\" __global__ void __launch_bounds__(" << out.block_size;
        if (opt.waves_per_simd) s << ", " << opt.waves_per_simd;
        s << ")\n" << out.kernel_name << (which == entry::converge ? "_converge" : which == entry::max ? "_max" : which == entry::batch ? "_batch" : "") << "(";
        for (size_t i = 0; i < it.symbols.size(); i++) {
            s << (out.input_written[i] ? "" : "const ") << "real *__restrict__ in" << i << ", ";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "real *__restrict__ out" << o << ", ";
        }
        for (size_t p = 0; p < out.packs.size(); p++) {
            s << "const real *__restrict__ pack" << p << ", ";
        }
        if (it.has_random()) s << "gf_mt_state *__restrict__ random_states, ";
        s << "unsigned int *__restrict__ flags, const unsigned long long n, ";
        if (piece.role == piece_role::middle) s << "unsigned char *__restrict__ flagged, ";
        if (piece.role == piece_role::last) {
            s << "unsigned char *__restrict__ flagged, unsigned int *__restrict__ redo_list, unsigned int *__restrict__ redo_count, "
              << "const unsigned int first, ";
        }
        if (piece.role == piece_role::redo) {
            s << "unsigned char *__restrict__ flagged, const unsigned int *__restrict__ redo_list, unsigned int *__restrict__ redo_count, ";
        }
        if (which == entry::converge) {
            s << "const real tolerance,\n        const unsigned int max_iterations, unsigned int *__restrict__ iterations) {\n";
        } else if (which == entry::batch) {
            s << "const unsigned int steps,\n        unsigned long long *__restrict__ reduce, const unsigned int *__restrict__ stop";
            for (size_t k = 0; k < it.setters.size(); k++) s << ", real *__restrict__ undo" << k;
            s << ") {\n    if (stop && *stop) return;\n";
        } else if (which == entry::max) {
            s << "const unsigned int steps,\n        unsigned long long *__restrict__ reduce, const unsigned int *__restrict__ stop) {\n"
//  A converge loop enqueues its passes ahead of the host (gf_hip.cpp): once the loop's test has
//  come out false on the device, the passes still in the queue must not touch the state.
              << "    if (stop && *stop) return;\n";
        } else {
            s << "const unsigned int steps) {\n";
        }
    }

//  LDS: the staged packs (copied once per workgroup) and the parking slots.
    void lds_setup() {
        if (lds_used) {
            s << "    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];\n";
            size_t offset = 0;
            for (size_t p = 0; p < out.packs.size(); p++) {
                if (!out.packs[p].in_lds) continue;
                const size_t count = out.packs[p].elements();
                s << "    real *lds" << p << " = reinterpret_cast<real *> (lds_raw + " << offset << ");\n";
//  (the redo kernel runs after every pass and nearly always finds its list empty: it stages nothing then)
                s << "    " << (piece.role == piece_role::redo ? "if (*redo_count != 0u) " : "")
                  << "for (unsigned int k = threadIdx.x; k < " << count << "u; k += blockDim.x) lds" << p
                  << "[k] = pack" << p << "[k];\n";
                offset += (count*esize + 15)/16*16;
            }
            s << "    __syncthreads();\n";
            if (!asm_statement.empty()) {
//  The assembly body addresses LDS itself: the staged packs' bases and the lane's slot 0 as 32-bit LDS addresses.
                size_t base = 0;
                for (size_t p = 0; p < out.packs.size(); p++) {
                    if (!out.packs[p].in_lds) continue;
                    s << "    park_t *lds_address" << p << " = (park_t *)(lds_raw + " << base << ");\n";
                    base += (out.packs[p].elements()*esize + 15)/16*16;
                }
                const uint32_t per_base = 65536u/(out.block_size*8u);
                for (uint32_t k = 0; k*per_base < park_slots; k++) {
                    s << "    park_t *park" << k << " = (park_t *)(lds_raw + " << park_offset + static_cast<size_t> (k)*65536u << ") + threadIdx.x;\n";
                }
            } else
            if (park_slots) {
//  Explicit LDS address space so the accesses stay ds_write_b64/ds_read_b64.
//  Two laundered copies of the same LDS pointer: the compiler cannot prove that a read through
//  `park_read` aliases a write through `park` (so no store-to-load forwarding, which would put
//  the value back in a register) nor that it does not (so a read is never hoisted above an
//  earlier write).  Not volatile: waits are placed at the first use, not after the read.
                s << "    park_t *park = (park_t *)(lds_raw + " << park_offset << ") + threadIdx.x;\n";
                s << "    park_t *park_read = park;\n";
                s << "    asm volatile(\"\" : \"+v\"(park));\n";
                s << "    asm volatile(\"\" : \"+v\"(park_read));\n";
            }
        }
    }

//  The grid-stride loop over tiles and the loads of a tile's state.
    void tile_open(const entry which) {
        if (which == entry::max) {
            s << "    real lane_max = -__builtin_huge_val" << sfx << "();\n";
        }
        if (which == entry::batch) {
            for (uint32_t b = 0; b < opt.converge_batch; b++) {
                s << "    real lane_max" << b << " = -__builtin_huge_val" << sfx << "();\n";
            }
        }
        if (it.has_random()) {
//  cuda_context.hpp:509-522, :817: thread t of the (sequentially launched) blocks owns state t and
//  draws for elements t, t + 1024, ... in that order; the launch has at most 1024 lanes (gf_hip.cpp).
            s << "    gf_mt_state &random_state = random_states[blockIdx.x*blockDim.x + threadIdx.x];\n";
        }
        if (piece.role == piece_role::redo) {
//  The rays of the redo list (ensemble indices); nothing to do, and nothing read but the count, when it is empty.
            s << "    const unsigned int redo_total = *redo_count;\n"
              << "    for (unsigned long long j = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x; j < redo_total;\n"
              << "         j += gridDim.x*static_cast<unsigned long long> (blockDim.x)) {\n"
              << "        const unsigned long long i = redo_list[j];\n"
              << "        if (i >= n) continue;\n"
              << "        flagged[i] = 0;\n";
        } else {
            s << "    for (unsigned long long i = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x; i < n;\n"
              << "         i += gridDim.x*static_cast<unsigned long long> (blockDim.x)) {\n";
        }
        if (piece.role == piece_role::last) s << "        bool redo = flagged[i] != 0;\n";
        for (size_t i = 0; i < it.symbols.size(); i++) {
            std::string symbol = it.symbols[i];
            for (auto &ch : symbol) {
                if (ch == '\\' || ch == '\n' || ch == '\r') ch = ' ';
            }
            if ((opt.nontemporal == 1 || opt.nontemporal == 3) && !cx) {
                s << "        real v" << i << " = __builtin_nontemporal_load(in" << i << " + i);  // " << symbol << "\n";
            } else {
                s << "        real v" << i << " = in" << i << "[i];  // " << symbol << "\n";
            }
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "        real o" << o << " = " << value_literal(0.0) << ";\n";
        }
        if (which == entry::batch) {
            for (size_t k = 0; k < it.setters.size(); k++) {
                s << "        __builtin_nontemporal_store(v" << it.setters[k].input << ", undo" << k << " + i);\n";
            }
        }
    }

//  One pass over the tile: the per-ray converge loop, or `steps` passes of the body.
    void pass_open(const entry which) {
        if (which == entry::converge) {
            s << "        unsigned int count = 0;\n"
              << "        bool active = true;\n"
              << "        real last_max = " << (f64 ? "__DBL_MAX__" : "__FLT_MAX__") << ", off_last_max = last_max;\n"
              << "        for (;;) {\n"
              << "            if (active) {\n";
        } else if (which == entry::batch) {
            s << "        if (" << batch_pass << "u < steps) {\n";
            s << "            {\n";
        } else {
            s << "        for (unsigned int step = 0; step < steps; step++) {\n";
            s << "            {\n";
        }
        for (size_t k = 0; k < it.setters.size(); k++) {
            s << "            real sv" << k << ";\n";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "            real so" << o << ";\n";
        }
    }

//  The body with its checks and its IEEE second body, the write-back into the lane's state,
//  the end of the pass loop.
    void pass(const entry which) {
        if (use_shared && fast) {
//  Tolerance mode (GFHIP_DIVISION=fast): the shared-reciprocal body, nothing to check, nothing to redo.
            s << "            {\n";
            body(true);
            s << "            }\n";
        } else if (use_shared) {
            s << "            bool bad = false, zero = false;\n";
            s << "            {\n";
            s << "                float dmax = gf_magnitude(" << literal(1.0) << "), dmin = dmax;   // extreme |denominator| of this pass\n";
            if (f64) s << "                float vmax = dmax;                                   // extreme |stored value|, |index quotient|\n";
            bool sqrt_window = false;
            if (!f64 && opt.window_sqrt_f32) {
                for (auto &c : it.code) sqrt_window = sqrt_window || c.op == GFIR_SQRT;
            }
            if (sqrt_window) s << "                float smax = 1.0f, smin = 1.0f;                       // extreme sqrtf argument\n";
            if (f64 && !fixup) s << "                float zmin = __builtin_inff();                       // smallest |value| whose zero would be observed\n";
            if (track_numerators) s << "                unsigned int nmin = 0xFFFFFFFFu;                    // smallest non-zero |numerator| key\n";
            if (asm_statement.empty()) body(true); else s << asm_statement;
//  The finite checks run on the same fp32 image as the window check (a non-finite value, or a
//  double of 2^1017 and more, reads as a float NaN/infinity and fails the comparison): half a
//  v_maximum3_f32 per value.
            std::vector<std::string> quotient_results;
            for (size_t k = 0; f64 && k < it.setters.size(); k++) {
                s << "                vmax = __builtin_elementwise_maximum(vmax, gf_magnitude(sv" << k << "));\n";
                if (after_division[it.setters[k].value]) quotient_results.push_back("sv" + std::to_string(k));
            }
            for (size_t o = 0; f64 && o < it.outputs.size(); o++) {
//  A handed-over value is not a stored value: a non-finite or zero one is seen where it ends up.
                if (o < piece.output_handed_over.size() && piece.output_handed_over[o]) continue;
                s << "                vmax = __builtin_elementwise_maximum(vmax, gf_magnitude(so" << o << "));\n";
                if (after_division[it.outputs[o]]) quotient_results.push_back("so" + std::to_string(o));
            }
//  Windows: see the contract at the top of prelude.hpp.  fp32 (quotients through fp64): every
//  denominator finite and non-zero, nothing else.
            const char *low = f64 ? "0x1p-500" : "0x1p-149f";
            const char *high = f64 ? "0x1p+500" : "0x1.fffffep+127f";
            s << "                bad = " << (f64 ? "!(vmax < __builtin_inff()) || " : "") << "!(dmin >= gf_magnitude(" << low
              << ")) || !(dmax <= gf_magnitude(" << high << "))";
            if (track_numerators) s << " || nmin < gf_numerator_key(0x1p-450)";     // fp64 only (track_numerators)
            if (sqrt_window) s << " || !(smin >= 0x1p-96f) || !(smax < __builtin_inff())";
            s << ";\n";
//  With v_div_fixup in the quotients the sign of a zero is the IEEE one and stored zeros need no
//  second look (GFHIP_DIV_FIXUP=1: for ensembles that keep exact zeros in their state, e.g. a
//  symmetry plane, where this check would send every pass through the IEEE function).
//  (the image of a double below 2^-1042 is zero as well: such a value takes the IEEE function too)
            if (f64 && !fixup) {
                for (auto &value : quotient_results) {
                    s << "                zmin = __builtin_elementwise_minimum(zmin, gf_magnitude(" << value << "));\n";
                }
                s << "                zero = zmin == 0.0f;\n";
            }
            s << "            }\n";
//  Lanes that failed a check redo the pass with the compiler's IEEE division (the inputs of the
//  pass are still in v*).  Status bit 0: a window/finite check failed; bit 1: a stored zero that
//  came from a quotient.  A bit is set once: lanes that find it set only read it (an atomic per
//  flagged lane on one address serialises at ~11 ns each).
            if (piece.role == piece_role::middle || piece.role == piece_role::last) {
//  No IEEE function in this kernel: the lane is redone by the redo launch (segments.hpp).
                s << "            if (__builtin_expect(bad || zero, 0)) {\n"
                  << "                const unsigned int why = bad ? 1u : 2u;\n"
                  << "                if ((__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & why) == 0u) atomicOr(flags, why);\n"
                  << (piece.role == piece_role::middle ? "                flagged[i] = 1;\n" : "                redo = true;\n")
                  << "            }\n";
            } else {
            s << "            if (__builtin_expect(bad || zero, 0)) {\n"
              << "                const unsigned int why = bad ? 1u : 2u;\n"
              << "                if ((__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & why) == 0u) atomicOr(flags, why);\n"
              << "                const " << out.kernel_name << "_results redo = " << out.kernel_name << "_ieee(";
            for (size_t i = 0; i < it.symbols.size(); i++) s << "v" << i << ", ";
            for (size_t p = 0; p < out.packs.size(); p++) s << (out.packs[p].in_lds ? "lds" : "pack") << p << ", ";
            for (size_t i = 0; i < it.symbols.size(); i++) {
                if (it.indexed_length(static_cast<uint32_t> (i))) s << "in" << i << ", ";
            }
            if (park_slots) s << "park, park_read, ";
            s << "0);\n";
            for (size_t k = 0; k < it.setters.size(); k++) s << "                sv" << k << " = redo.sv" << k << ";\n";
            for (size_t o = 0; o < it.outputs.size(); o++) s << "                so" << o << " = redo.so" << o << ";\n";
            s << "            }\n";
            }
        } else {
            s << "            {\n";
            body(false);
            s << "            }\n";
        }
//  SAFE_MATH stores isnan(x) ? 0 : x (per part for complex values), cpu_context.hpp:530-547.
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "            o" << o << " = " << (safe ? "gf_nan_to_zero(so" + std::to_string(o) + ")" : "so" + std::to_string(o)) << ";\n";
        }
        for (size_t k = 0; k < it.setters.size(); k++) {
            s << "            v" << it.setters[k].input << " = " << (safe ? "gf_nan_to_zero(sv" + std::to_string(k) + ")" : "sv" + std::to_string(k)) << ";\n";
        }
        if (which == entry::converge) {
//  converge_item::run for this ray:  while (A && B && C && iterations++ < max) {...}
            const std::string fabs_ = std::string("__builtin_fabs") + (f64 ? "" : "f");
            s << "            const real residual = so" << it.outputs.size() - 1 << ";\n"
              << "            bool go = " << fabs_ << "(residual) > " << fabs_ << "(tolerance) &&\n"
              << "                      " << fabs_ << "(last_max - residual) > " << fabs_ << "(tolerance) &&\n"
              << "                      " << fabs_ << "(off_last_max - residual) > " << fabs_ << "(tolerance);\n"
              << "            if (go) { go = count < max_iterations; count++; }\n"
              << "            if (go) { last_max = residual; if (!(count%2u)) off_last_max = residual; }\n"
              << "            active = go;\n"
              << "            }\n"
              << "            if (__ballot(active) == 0ull) break;   // the whole wavefront has stalled\n"
              << "        }\n"
              << "        atomicMax(iterations, count);\n";
        } else if (which == entry::batch) {
//  The max of this pass's last output, std::max_element's way (see the `_max` epilogue in stores()).
            const std::string last = "o" + std::to_string(it.outputs.size() - 1);
            const std::string lane = "lane_max" + std::to_string(batch_pass);
            s << "            " << lane << " = " << last << " > " << lane << " ? " << last << " : " << lane << ";\n";
            s << "            if (i == 0ull && " << last << " != " << last << ") " << lane << " = " << last << ";\n";
            s << "            }\n";
            s << "        }\n";
        } else {
            s << "            }\n";
            s << "        }\n";
        }
    }

    void stores(const entry which) {
        if (piece.role == piece_role::last) {
//  Lanes to redo leave their state untouched and line up in the redo list: one atomicAdd per
//  wavefront that has any, the lanes of the wavefront take consecutive places.
            s << "        {\n"
              << "            const unsigned long long redo_lanes = __ballot(redo);\n"
              << "            if (redo_lanes) {\n"
              << "                const unsigned int lane = threadIdx.x & 63u;\n"
              << "                const unsigned int leader = static_cast<unsigned int> (__builtin_ctzll(redo_lanes));\n"
              << "                unsigned int base = 0;\n"
              << "                if (lane == leader) base = atomicAdd(redo_count, static_cast<unsigned int> (__builtin_popcountll(redo_lanes)));\n"
              << "                base = __shfl(base, static_cast<int> (leader), 64);\n"
              << "                if (redo) redo_list[base + static_cast<unsigned int> (__builtin_popcountll(redo_lanes & ((1ull << lane) - 1ull)))] = first + static_cast<unsigned int> (i);\n"
              << "            }\n"
              << "        }\n"
              << "        if (!redo) {\n";
        }
//  Stores: setters first, then outputs (cpu_context.hpp:522-580).
        for (size_t i = 0; i < it.symbols.size(); i++) {
            if (!out.input_written[i]) continue;
            if ((opt.nontemporal == 1 || opt.nontemporal == 2) && !cx) {
                s << "        __builtin_nontemporal_store(v" << i << ", in" << i << " + i);\n";
            } else {
                s << "        in" << i << "[i] = v" << i << ";\n";
            }
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            if ((opt.nontemporal == 1 || opt.nontemporal == 2) && !cx) {
                s << "        __builtin_nontemporal_store(o" << o << ", out" << o << " + i);\n";
            } else {
                s << "        out" << o << "[i] = o" << o << ";\n";
            }
        }
        if (piece.role == piece_role::last) s << "        }\n";
        if (which == entry::max) {
//  The max of the last output (create_max_call's argument) as cpu_context takes it — std::max_element,
//  cpu_context.hpp:306-322: a NaN is never selected, unless it is element 0, which then stays the
//  maximum (every `max < x` is false).  The lane that owns element 0 keeps such a NaN.
            const std::string last = "o" + std::to_string(it.outputs.size() - 1);
            s << "        lane_max = " << last << " > lane_max ? " << last << " : lane_max;\n";
            s << "        if (i == 0ull && " << last << " != " << last << ") lane_max = " << last << ";\n";
        }
        s << "    }\n";
        if (which == entry::max) {
            max_epilogue("lane_max", "reduce", "wave_max");
        }
        if (which == entry::batch) {
            for (uint32_t b = 0; b < opt.converge_batch; b++) {
                s << "    if (" << b << "u < steps) {\n";
                max_epilogue("lane_max" + std::to_string(b), "(reduce + " + std::to_string(b) + ")", "wave_max" + std::to_string(b));
                s << "    }\n";
            }
        }
        if (piece.role == piece_role::redo) {
//  The list is empty again for the next pass: the workgroup that finishes last — every workgroup has read the count by
//  then — clears it and the arrival counter next to it (a memset per pass on the host's stream cost 5 us of every step).
            s << "    __syncthreads();\n"
              << "    if (threadIdx.x == 0u) {\n"
              << "        __threadfence();\n"
              << "        if (atomicAdd(redo_count + 1, 1u) == gridDim.x - 1u) {\n"
              << "            redo_count[0] = 0u;\n"
              << "            redo_count[1] = 0u;\n"
              << "        }\n"
              << "    }\n";
        }
        s << "}\n";
    }

//  Epilogue of create_max_call: 64-lane shuffle reduction, one LDS word per wave, ONE
//  device-scope atomicMax per workgroup on an order-preserving integer image of the value
//  (max is exact and order independent: the same bits as a serial scan).
    void max_epilogue(const std::string &lane, const std::string &target, const std::string &scratch) {
        const char *bits = f64 ? "unsigned long long" : "unsigned int";
        s << "    for (int offset = 32; offset > 0; offset >>= 1) {\n"
          << "        const real other = __shfl_down(" << lane << ", offset, 64);\n"
          << "        " << lane << " = other > " << lane << " ? other : " << lane << ";\n"
          << "    }\n"
          << "    __shared__ real " << scratch << "[" << out.block_size/64 << "];\n"
          << "    if ((threadIdx.x & 63u) == 0u) " << scratch << "[threadIdx.x >> 6] = " << lane << ";\n"
          << "    __syncthreads();\n"
          << "    if (threadIdx.x == 0u) {\n"
          << "        real block_max = " << scratch << "[0];\n"
          << "        for (unsigned int w = 1; w < (blockDim.x >> 6); w++) block_max = " << scratch << "[w] > block_max ? " << scratch << "[w] : block_max;\n"
          << "        const " << bits << " b = __builtin_bit_cast(" << bits << ", block_max);\n"
          << "        const " << bits << " top = static_cast<" << bits << "> (1) << " << (f64 ? 63 : 31) << ";\n"
          << "        atomicMax(" << target << ", static_cast<unsigned long long> (block_max != block_max ? static_cast<" << bits << "> (~static_cast<" << bits << "> (0))\n"
          << "                                                             : (b & top) ? static_cast<" << bits << "> (~b) : (b | top)));\n"
          << "    }\n";
    }
};

//------------------------------------------------------------------------------
///  @brief Lower one item.
//------------------------------------------------------------------------------
inline lowered lower(const item &original, const codegen_options &opt = codegen_options::from_environment(),
                     const piece_info &piece = piece_info()) {
    item scheduled;
//  (the assembly body keeps the order of a piece that is already in emission order: its annotations number the
//  records of the piece that gfhip_export_piece hands out, tests/asm_symbolic.py)
    if (opt.schedule_for_pressure && !(opt.asm_body && piece.scheduled)) {
        scheduled = schedule_for_pressure(original);
    }
    const item &it = scheduled.code.empty() ? original : scheduled;
    lowered out;
    const size_t esize = it.element_size();

//  Which inputs are overwritten.
    out.input_written.assign(it.symbols.size(), false);
    for (auto &s : it.setters) {
        out.input_written[s.input] = true;
    }

    const table_layout layout = layout_tables(it, opt);
    const std::vector<int> &parent = layout.parent;
    const std::vector<double> &factor = layout.factor;
    const std::vector<uint32_t> &table_pack = layout.table_pack, &table_column = layout.table_column;
    out.table_parent = parent;
    out.table_factor = factor;
    out.packs = layout.packs;
    size_t lds_used = layout.lds_used;
    out.block_size = opt.block_size;

//  Complex values, SAFE_MATH guards and random draws are off the hot path: no LDS parking, no
//  shared reciprocals (guarded divisions may divide by zero on purpose), and the max of a complex
//  output is the element of largest modulus, which reduce.hip finds.
    const bool generic = it.is_complex() || it.safe_math() || it.has_random();
    codegen_options plain = opt;
    if (generic) plain.park_in_lds = false;
//  The body as assembly (asm_body.hpp): only as the `last` piece of an item whose out-of-window lanes are redone by a
//  separate launch — an IEEE function compiled into the kernel would claim registers beyond the 256 the pool ends at.
    asm_body_text assembly;
    const bool divides_at_all = std::any_of(it.code.begin(), it.code.end(), [] (const gfir_instruction &c) { return c.op == GFIR_DIV; });
    if (opt.asm_body && piece.role == piece_role::last && !generic && divides_at_all && opt.division == division_mode::shared &&
        it.code.size() >= opt.asm_min_nodes) {
        const size_t per_block = 160u*1024u/opt.asm_waves;  // a workgroup of 256 lanes is one wave per SIMD of its CU
        const uint32_t slot_limit = lds_used < per_block ? static_cast<uint32_t> ((per_block - lds_used)/(static_cast<size_t> (opt.block_size)*esize)) : 0;
        asm_body_writer writer(it, opt, out.packs, parent, factor, table_pack, table_column, opt.block_size, slot_limit);
        assembly = writer.write();
        if (std::getenv("GFHIP_ASM_REPORT")) {
            std::fprintf(stderr, "assembly body of %s: %s; %zu vector, %zu scalar, %zu table loads, %zu LDS reads, %zu LDS writes, %zu waits, %u slots\n",
                         it.name.c_str(), assembly.ok ? "ok" : assembly.why.c_str(), assembly.vector, assembly.scalar, assembly.loads,
                         assembly.lds_reads, assembly.lds_writes, assembly.waits, assembly.slots);
        }
    }
    if (assembly.ok) plain.park_in_lds = false;
    uint32_t park_slots = 0;
    const std::vector<park_plan> plan = plain.park_in_lds && plain.park_capacity && piece.role != piece_role::redo
                                      ? plan_parking_belady(it, plain, lds_used, esize, park_slots)
                                      : plan_parking(it, plain, lds_used, esize, park_slots);
    if (assembly.ok) park_slots = assembly.slots;
    const size_t park_offset = lds_used;
    lds_used += static_cast<size_t> (park_slots)*opt.block_size*esize;
    out.lds_bytes = lds_used;
    out.park_slots = park_slots;

//  Nodes that depend on the result of a division (a stored zero among them may carry the wrong
//  sign without v_div_fixup), and whether the item divides at all.
    std::vector<bool> after_division(it.code.size(), false);
    bool divides = false;
    for (size_t i = 0; i < it.code.size(); i++) {
        const gfir_instruction &c = it.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        bool dependent = c.op == GFIR_DIV;
        if (c.op == GFIR_INPUT && c.a < piece.symbol_after_division.size() && piece.symbol_after_division[c.a]) dependent = true;
        divides = divides || c.op == GFIR_DIV;
        for (int k = 0; k < operand_count(c.op) && !dependent; k++) dependent = after_division[operands[k]];
        after_division[i] = dependent;
    }

    std::ostringstream s;
    out.kernel_name = "gfhip_" + it.name;
    emit_prelude(s, it, opt, out.packs.size());
//  Items without a division node need neither the checks nor the second body (their gather
//  indices then divide by the literal scale).
//  A segment that stores a quotient of an EARLIER segment has a zero to look at even if it divides nothing itself.
    bool stores_quotient = false;
    for (auto &st : it.setters) stores_quotient = stores_quotient || after_division[st.value];
    for (size_t o = 0; o < it.outputs.size(); o++) {
        const bool handed_over = o < piece.output_handed_over.size() && piece.output_handed_over[o];
        stores_quotient = stores_quotient || (!handed_over && after_division[it.outputs[o]]);
    }
    const bool checks_needed = divides || (piece.role != piece_role::none && stores_quotient);
    const bool use_shared = opt.division != division_mode::ieee && checks_needed && !generic && piece.role != piece_role::redo;
//  Entry points: `<name>` runs `steps` passes.  Small items with an output also get `<name>_max`
//  (the same, plus the max of the last output reduced inside the launch: create_max_call) and,
//  with a setter, `<name>_converge`, which runs the stall loop of workflow.hpp:179-205 PER RAY
//  inside the launch — every lane iterates on its own residual, a wavefront leaves the loop when
//  the ballot of still-active lanes is empty.  That is the reference's converge loop applied to
//  each ray as its own shard; it equals the reference's global-max loop when the rays are
//  identical (the benchmark) and is offered as gfhip_converge_per_ray.
    const bool small = it.code.size() <= 1500 && piece.role == piece_role::none;
    out.has_max = !it.outputs.empty() && small && !it.is_complex() && !it.has_random();
    out.has_converge = out.has_max && !it.setters.empty();
    codegen_options resolved = opt;
    if (resolved.nontemporal < 0) resolved.nontemporal = it.code.size() >= 100 ? 1 : 0;
    const std::string no_assembly;
    const bool as_assembly = assembly.ok && use_shared;
    if (as_assembly) resolved.waves_per_simd = opt.asm_waves;
    out.assembly = as_assembly;
    kernel_writer writer{s, it, resolved, out, parent, factor, table_pack, table_column, plan, lds_used, park_offset, park_slots,
                         use_shared, after_division, piece, as_assembly ? assembly.statement : no_assembly};
    if (park_slots || as_assembly) s << "typedef __attribute__((address_space(3))) real park_t;\n";
    if (use_shared && piece.role == piece_role::none && opt.division != division_mode::fast) writer.ieee_function();
    writer.kernel(entry::plain);
    if (out.has_max) writer.kernel(entry::max);
    if (out.has_converge) writer.kernel(entry::converge);
    if (out.has_converge && resolved.converge_batch > 1) {
        out.batch = resolved.converge_batch;
        writer.kernel(entry::batch);
    }

    out.source = s.str();
    out.hash = fnv1a(out.source + "|" + compile_flags());
    return out;
}

}  // namespace gfhip

#endif /* gfhip_codegen_hpp */

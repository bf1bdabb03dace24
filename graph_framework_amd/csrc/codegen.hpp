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
///  * packs that fit the LDS budget (the 1-D profile tables: 138 bins x 45
///    tables x 8 B = 50 KB) are staged into LDS once per workgroup; the 2-D
///    psi pack (1.5 MB) stays L2 resident (4 MiB per XCD);
///  * `steps` passes can run inside one launch with the state kept in
///    registers (the reference launches once per step, solver.hpp:382).
///
///  Pieces: options.hpp (knobs, cache hash), schedule.hpp (emission order),
///  tables.hpp (compaction, packs, LDS staging), parking.hpp (values that wait
///  in LDS), prelude.hpp (device helpers: division, window check, pow), and
///  lower() below, which writes the kernel text.
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

#include <fstream>

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
    uint32_t elements = 1;              ///< consecutive rays owned by one lane
    bool has_converge = false;          ///< the module also holds `<name>_converge`
    uint64_t hash = 0;
};

//------------------------------------------------------------------------------
///  @brief Writes the kernel text of one lowered item (`<name>` and, for items
///  that can converge per ray, `<name>_converge`).
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
    const uint32_t E;                               ///< rays per lane
    const bool packed;                              ///< ... held as one float2
    const bool use_shared;                          ///< shared-reciprocal division

    const bool f64 = it.dtype == GFIR_F64;
    const char *real = f64 ? "double" : "float";
    const std::string sfx = f64 ? "" : "f";
    const size_t esize = it.element_size();
    const size_t node_count = it.code.size();
    const std::string VT = packed ? "real2" : "real";         ///< type of a value of the pass
    bool prefetch = false;                          ///< next-tile prefetch in the kernel being written

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

//  The node-for-node body.  `shared` = divisions through a reciprocal shared by all
//  divisions with the same denominator (see gf_rcp/gf_div in the prelude); otherwise the
//  compiler's IEEE division.
    void body(const bool shared) {
        typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, double, double, double, double> group_key;
        std::map<group_key, std::string> groups;
        std::map<uint32_t, bool> reciprocal_done;
        std::set<std::string> coefficients;
        size_t group_count = 0;
//  Current name of every value (changes when a parked value is reloaded).
        std::vector<std::string> name(node_count);
        for (size_t v = 0; v < node_count; v++) name[v] = "r" + std::to_string(v);
        auto name_of = [&] (const uint32_t v) -> std::string { return name[v]; };
        auto index_expression = [&] (const uint32_t arg, const double scale, const double offset,
                                     const uint32_t length) -> std::string {
            std::ostringstream e;
            if (packed) {
//  Clamped quotient as a float2; the caller converts each component.
                const std::string zero = "(real2)(" + literal(0.0) + ")";
                const std::string top = "(real2)(" + literal(static_cast<double> (length - 1)) + ")";
                e << "__builtin_elementwise_min(__builtin_elementwise_max(";
                if (shared) {
                    e << "gf_div(" << name_of(arg) << " - (real2)(" << literal(offset) << "), (real2)(" << literal(scale)
                      << "), (real2)(" << literal(static_cast<double> (1.0f/static_cast<float> (scale))) << "))";
                } else {
                    e << "(" << name_of(arg) << " - (real2)(" << literal(offset) << "))/(real2)(" << literal(scale) << ")";
                }
                e << ", " << zero << "), " << top << ")";
                return e.str();
            }
            e << "static_cast<unsigned int> (__builtin_fmin" << sfx << "(__builtin_fmax" << sfx << "(";
            if (shared) {
//  The reciprocal literal is the correctly rounded 1/scale; gf_div's residual step makes
//  the quotient the correctly rounded (r - offset)/scale.
                e << "gf_div(" << name_of(arg) << " - " << literal(offset) << ", " << literal(scale) << ", "
                  << literal(f64 ? 1.0/scale : static_cast<double> (1.0f/static_cast<float> (scale))) << ")";
            } else {
                e << "(" << name_of(arg) << " - " << literal(offset) << ")/" << literal(scale);
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
        auto N = [&] (const uint32_t v) -> const std::string & { return name[v]; };
//  Value of table `t` at the cell of index group `group`: a load for stored tables, an exact
//  multiple of the parent's value otherwise; one definition per (group, table).
        std::function<std::string(const std::string &, uint32_t)> value_at =
            [&] (const std::string &group, const uint32_t t) -> std::string {
            const std::string value_name = "c" + group.substr(1) + "_" + std::to_string(t);
            if (coefficients.insert(value_name).second) {
                if (parent[t] >= 0) {
                    const std::string from = value_at(group, static_cast<uint32_t> (parent[t]));
                    s << ind << "const " << VT << " " << value_name << " = " << literal(factor[t]) << "*" << from << ";\n";
                } else {
                    const uint32_t pi = table_pack[t];
                    const std::string base = (out.packs[pi].in_lds ? "lds" : "pack") + std::to_string(pi);
                    if (packed) {
                        s << ind << "const real2 " << value_name << " = {" << base << "[" << group << "_0 + " << table_column[t]
                          << "u], " << base << "[" << group << "_1 + " << table_column[t] << "u]};\n";
                    } else {
                        s << ind << "const real " << value_name << " = " << base << "[" << group << " + " << table_column[t] << "u];\n";
                    }
                }
            }
            return value_name;
        };
        std::map<uint32_t, std::pair<std::string, uint32_t>> deferred;     // gather node -> (group, table)
        auto define = [&] (const uint32_t v) {
            auto found = deferred.find(v);
            if (found == deferred.end()) return;
            const std::string value = value_at(found->second.first, found->second.second);
            s << ind << "const " << VT << " r" << v << " = " << value << ";\n";
            deferred.erase(found);
        };
//  Next-tile prefetch: vmcnt retires in order, so a load issued before a gather makes the
//  gather's wait last as long as the (HBM-latency) prefetch.  The prefetch goes after the last
//  gather of the pass; the rest of the pass (>= ~20 % of it in the RK4 item) covers its latency.
        size_t prefetch_position = 0;
        {
//  The latest gather that is followed by at least `prefetch_min_gap` gather-free nodes (the
//  tail of the pass counts as a gap); failing that, the one followed by the widest gap.
            std::vector<size_t> gathers;
            for (size_t i = 0; i < it.code.size(); i++) {
                if (it.code[i].op == GFIR_GATHER1 || it.code[i].op == GFIR_GATHER2) gathers.push_back(i);
            }
            size_t widest = 0;
            bool satisfied = false;
            for (size_t k = 0; k < gathers.size(); k++) {
                const size_t next = k + 1 < gathers.size() ? gathers[k + 1] : it.code.size();
                const size_t gap = next - gathers[k] - 1;
                if (gap >= opt.prefetch_min_gap) {
                    if (!(opt.pipeline_tiles && satisfied)) prefetch_position = gathers[k] + 1;
                    satisfied = true;
                } else if (!satisfied && gap > widest) {
                    widest = gap;
                    prefetch_position = gathers[k] + 1;
                }
            }
        }
        for (size_t i = 0; i < it.code.size(); i++) {
            const gfir_instruction &c = it.code[i];
            if (prefetch && i == prefetch_position) {
//  Unconditional (a branch would split the scheduling region): passes before the last one of a
//  fused launch, and the last tile, re-read this tile's own (cached) element.
                s << ind << "unsigned long long ahead = (step + 1u == steps && g + stride < groups) ? g + stride : g;\n";
                if (prefetch_position > 0) {
                    define(static_cast<uint32_t> (prefetch_position - 1));
//  Tie the address to the last gather's result, or the compiler hoists the loads to the top.
                    s << ind << "asm volatile(\"\" : \"+v\"(ahead) : \"v\"(" << N(static_cast<uint32_t> (prefetch_position - 1)) << "));\n";
                }
                for (size_t k = 0; k < it.symbols.size(); k++) {
                    s << ind << "next" << k << " = in" << k << "[ahead];\n";
                }
                if (opt.pipeline_tiles) {
//  The previous tile's results leave here: a store issued at the end of a pass would still be
//  in flight at the next pass's first gather wait (vmcnt retires stores and loads in order).
                    s << ind << "if (step == 0u && have_pending) {\n";
                    for (size_t k = 0; k < it.symbols.size(); k++) {
                        if (out.input_written[k]) s << ind << "    in" << k << "[pending_index] = pending_v" << k << ";\n";
                    }
                    for (size_t o = 0; o < it.outputs.size(); o++) {
                        s << ind << "    out" << o << "[pending_index] = pending_o" << o << ";\n";
                    }
                    s << ind << "    have_pending = false;\n";
                    s << ind << "}\n";
                }
//  ... and keep the scheduler from sinking them to the end of the pass.
                s << ind << "__builtin_amdgcn_sched_barrier(0);\n";
            }
            reload(i);
            {
                const uint32_t operands[3] = {c.a, c.b, c.c};
                for (int k = 0; k < operand_count(c.op); k++) define(operands[k]);
            }
            if ((opt.sched_barrier_every && i && i%opt.sched_barrier_every == 0) ||
                std::find(it.fences.begin(), it.fences.end(), static_cast<uint32_t> (i)) != it.fences.end()) {
                s << ind << "__builtin_amdgcn_sched_barrier(0);\n";
            }
            switch (c.op) {
                case GFIR_CONST:
                    s << ind << "const " << VT << " r" << i << " = " << literal(c.imm[0]) << ";\n";
                    break;
                case GFIR_INPUT:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? "V" + std::to_string(c.a) : "v" + std::to_string(c.a) + "[e]") << ";\n";
                    break;
                case GFIR_ADD:
                    s << ind << "const " << VT << " r" << i << " = " << N(c.a) << " + " << N(c.b) << ";\n";
                    break;
                case GFIR_SUB:
                    s << ind << "const " << VT << " r" << i << " = " << N(c.a) << " - " << N(c.b) << ";\n";
                    break;
                case GFIR_MUL:
                    s << ind << "const " << VT << " r" << i << " = " << N(c.a) << "*" << N(c.b) << ";\n";
                    break;
                case GFIR_DIV:
                    if (shared) {
                        if (!reciprocal_done[c.b]) {
                            reciprocal_done[c.b] = true;
                            s << ind << "const " << VT << " q" << c.b << " = gf_rcp(" << N(c.b) << ");\n";
                            for (const char *member : {".x", ".y"}) {
                                const std::string component = N(c.b) + (packed ? member : "");
                                s << ind << "dmax = __builtin_elementwise_maximum(dmax, gf_magnitude(" << component << "));\n";
                                s << ind << "dmin = __builtin_elementwise_minimum(dmin, gf_magnitude(" << component << "));\n";
                                if (!packed) break;
                            }
                        }
                        s << ind << "const " << VT << " r" << i << " = gf_div(" << N(c.a) << ", " << N(c.b) << ", q" << c.b << ");\n";
                    } else {
                        s << ind << "const " << VT << " r" << i << " = " << N(c.a) << "/" << N(c.b) << ";\n";
                    }
                    break;
                case GFIR_FMA:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_fma2") : "__builtin_fma" + std::string(sfx)) << "(" << N(c.a) << ", " << N(c.b)
                      << ", " << N(c.c) << ");\n";
                    break;
                case GFIR_SQRT:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_sqrt2") : "__builtin_sqrt" + std::string(sfx)) << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_POWI: {
                    s << ind << "const " << VT << " r" << i << " = " << N(c.a);
                    for (uint32_t k = 1; k < c.aux; k++) s << "*" << N(c.a);
                    s << ";\n";
                    break;
                }
                case GFIR_POW:
                    if (f64 && opt.pow_three_halves && it.code[c.b].op == GFIR_CONST && it.code[c.b].imm[0] == 1.5) {
                        s << ind << "const " << VT << " r" << i << " = gf_pow_three_halves(r" << c.a << ");\n";
                    } else {
                        s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_pow2") : "pow" + std::string(sfx)) << "(" << N(c.a) << ", " << N(c.b) << ");\n";
                    }
                    break;
                case GFIR_SIN:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_sin2") : "sin" + std::string(sfx)) << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_COS:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_cos2") : "cos" + std::string(sfx)) << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_ATAN2:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_atan22") : "atan2" + std::string(sfx)) << "(" << N(c.b) << ", " << N(c.a) << ");\n";
                    break;
                case GFIR_EXP:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_exp2") : "exp" + std::string(sfx)) << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_LOG:
                    s << ind << "const " << VT << " r" << i << " = " << (packed ? std::string("gf_log2") : "log" + std::string(sfx)) << "(" << N(c.a) << ");\n";
                    break;
                case GFIR_GATHER1:
                case GFIR_GATHER2: {
                    const table &t = it.tables[c.aux];
                    const bool two = c.op == GFIR_GATHER2;
                    const group_key key(c.a, two ? c.b : GFIR_NONE, t.rows, t.cols,
                                        c.imm[0], c.imm[1], two ? c.imm[2] : 0.0, two ? c.imm[3] : 0.0);
                    auto g = groups.find(key);
                    if (g == groups.end()) {
                        const std::string name = "g" + std::to_string(group_count++);
                        const pack &p = out.packs[table_pack[c.aux]];
                        if (packed) {
                            s << ind << "const real2 " << name << "_x = " << index_expression(c.a, c.imm[0], c.imm[1], two ? t.rows : t.cols) << ";\n";
                            if (two) {
                                s << ind << "const real2 " << name << "_y = " << index_expression(c.b, c.imm[2], c.imm[3], t.cols) << ";\n";
                            }
                            for (int component = 0; component < 2; component++) {
                                const char *member = component ? ".y" : ".x";
                                s << ind << "const unsigned int " << name << "_" << component << " = (static_cast<unsigned int> ("
                                  << name << "_x" << member << ")";
                                if (two) {
                                    s << "*" << t.cols << "u + static_cast<unsigned int> (" << name << "_y" << member << ")";
                                }
                                s << ")*" << p.stride << "u;\n";
                            }
                            g = groups.insert({key, name}).first;
                        } else {
                        s << ind << "const unsigned int " << name << " = (";
                        if (two) {
                            s << index_expression(c.a, c.imm[0], c.imm[1], t.rows) << "*" << t.cols << "u + "
                              << index_expression(c.b, c.imm[2], c.imm[3], t.cols);
                        } else {
                            s << index_expression(c.a, c.imm[0], c.imm[1], t.cols);
                        }
                        s << ")*" << p.stride << "u;\n";
                        g = groups.insert({key, name}).first;
                        }
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
                    s << ind << "const " << VT << " r" << i << " = " << value << ";\n";
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

    void kernel(const bool converge) {
        signature(converge);
        lds_setup();
        tile_open(converge);
        pass_open(converge);
        pass(converge);
        stores();
    }

//  Kernel name, arguments.
    void signature(const bool converge) {
        s << "extern \"C\" __global__ void __launch_bounds__(" << out.block_size;
        if (opt.waves_per_simd) s << ", " << opt.waves_per_simd;
        s << ")\n" << out.kernel_name << (converge ? "_converge" : "") << "(";
        for (size_t i = 0; i < it.symbols.size(); i++) {
            s << (out.input_written[i] ? "" : "const ") << "real *__restrict__ in" << i << ", ";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "real *__restrict__ out" << o << ", ";
        }
        for (size_t p = 0; p < out.packs.size(); p++) {
            s << "const real *__restrict__ pack" << p << ", ";
        }
        if (converge) {
            s << "unsigned int *__restrict__ flags, const unsigned long long n, const real tolerance,\n"
              << "        const unsigned int max_iterations, unsigned int *__restrict__ iterations) {\n";
        } else {
            s << "unsigned int *__restrict__ flags, const unsigned long long n, const unsigned int steps) {\n";
        }
    }

    void lds_setup() {
//  LDS staging.
        if (lds_used) {
            s << "    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];\n";
            size_t offset = 0;
            for (size_t p = 0; p < out.packs.size(); p++) {
                if (!out.packs[p].in_lds) continue;
                const size_t count = out.packs[p].elements();
                s << "    real *lds" << p << " = reinterpret_cast<real *> (lds_raw + " << offset << ");\n";
                s << "    for (unsigned int k = threadIdx.x; k < " << count << "u; k += blockDim.x) lds" << p
                  << "[k] = pack" << p << "[k];\n";
                offset += (count*esize + 15)/16*16;
            }
            s << "    __syncthreads();\n";
            if (park_slots) {
//  Explicit LDS address space so the accesses stay ds_write_b64/ds_read_b64.
//  Two laundered copies of the same LDS pointer: the compiler cannot prove that a read through
//  `park_read` aliases a write through `park` (so no store-to-load forwarding, which would put
//  the value back in a register) nor that it does not (so a read is never hoisted above an
//  earlier write).  Not volatile: waits are placed at the first use, not after the read.
                s << "    typedef __attribute__((address_space(3))) real park_t;\n";
                s << "    park_t *park = (park_t *)(lds_raw + " << park_offset << ") + threadIdx.x;\n";
                s << "    park_t *park_read = park;\n";
                s << "    asm volatile(\"\" : \"+v\"(park));\n";
                s << "    asm volatile(\"\" : \"+v\"(park_read));\n";
            }
        }
    }

//  The grid-stride loop over tiles and the loads of a tile's state.
    void tile_open(const bool converge) {
        if (E > 1) {
            s << "    typedef real vec_t __attribute__((ext_vector_type(" << E << ")));\n";
            s << "    const bool aligned = ((0";
            for (size_t i = 0; i < it.symbols.size(); i++) s << " | reinterpret_cast<unsigned long long> (in" << i << ")";
            for (size_t o = 0; o < it.outputs.size(); o++) s << " | reinterpret_cast<unsigned long long> (out" << o << ")";
            s << ") & " << (E*esize - 1) << "ull) == 0;\n";
        }
        s << "    const unsigned long long groups = (n + " << (E - 1) << "ull)/" << E << "ull;\n";
        prefetch = opt.prefetch_next_tile && E == 1 && !converge;
        if (prefetch) {
//  Software pipelining across grid-stride tiles: at one wave per SIMD nothing else hides the
//  HBM latency of a tile's first loads, so they are issued one tile ahead.
            s << "    const unsigned long long stride = gridDim.x*static_cast<unsigned long long> (blockDim.x);\n";
            s << "    unsigned long long g = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x;\n";
            for (size_t i = 0; i < it.symbols.size(); i++) {
                s << "    real next" << i << " = g < groups ? in" << i << "[g] : " << literal(0.0) << ";\n";
            }
            if (opt.pipeline_tiles) {
//  Results of the previous tile, stored one tile late (see the body).
                s << "    bool have_pending = false;\n    unsigned long long pending_index = 0;\n";
                for (size_t i = 0; i < it.symbols.size(); i++) {
                    if (out.input_written[i]) s << "    real pending_v" << i << " = " << literal(0.0) << ";\n";
                }
                for (size_t o = 0; o < it.outputs.size(); o++) {
                    s << "    real pending_o" << o << " = " << literal(0.0) << ";\n";
                }
            }
            s << "    for (; g < groups; g += stride) {\n";
        } else {
            s << "    for (unsigned long long g = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x; g < groups;\n"
              << "         g += gridDim.x*static_cast<unsigned long long> (blockDim.x)) {\n";
        }
        s << "        const unsigned long long i = g*" << E << "ull;\n";
        if (E > 1) {
            s << "        const bool full = aligned && i + " << E << "ull <= n;\n";
        }
        for (size_t i = 0; i < it.symbols.size(); i++) {
            std::string symbol = it.symbols[i];
            for (auto &ch : symbol) {
                if (ch == '\\' || ch == '\n') ch = ' ';
            }
            s << "        real v" << i << "[" << E << "];  // " << symbol << "\n";
            if (E > 1) {
//  (loaded below, all arrays under one branch)
            } else if (prefetch) {
                s << "        v" << i << "[0] = next" << i << ";\n";
            } else {
                s << "        v" << i << "[0] = in" << i << "[i];\n";
            }
        }
        if (E > 1) {
            s << "        if (full) {\n";
            for (size_t i = 0; i < it.symbols.size(); i++) {
                s << "            const vec_t t" << i << " = *reinterpret_cast<const vec_t *> (in" << i << " + i);\n";
            }
            for (size_t i = 0; i < it.symbols.size(); i++) {
                s << "            for (unsigned int e = 0; e < " << E << "u; e++) v" << i << "[e] = t" << i << "[e];\n";
            }
            s << "        } else {\n";
            for (size_t i = 0; i < it.symbols.size(); i++) {
                s << "            for (unsigned int e = 0; e < " << E << "u; e++) v" << i << "[e] = i + e < n ? in" << i << "[i + e] : "
                  << (packed ? "in" + std::to_string(i) + "[i]" : literal(0.0)) << ";\n";
            }
            s << "        }\n";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "        real o" << o << "[" << E << "] = {};\n";
        }
    }

//  One pass over the tile: the per-ray converge loop, or `steps` passes of the body.
    void pass_open(const bool converge) {
        if (converge) {
            s << "        unsigned int count = 0;\n"
              << "        bool active = true;\n"
              << "        real last_max = " << (f64 ? "__DBL_MAX__" : "__FLT_MAX__") << ", off_last_max = last_max;\n"
              << "        for (;;) {\n"
              << "            if (active) {\n"
              << "            const unsigned int e = 0;\n";
        } else if (packed) {
            for (size_t i = 0; i < it.symbols.size(); i++) {
                s << "        real2 V" << i << " = {v" << i << "[0], v" << i << "[1]};\n";
            }
            for (size_t o = 0; o < it.outputs.size(); o++) {
                s << "        real2 O" << o << " = {};\n";
            }
            s << "        for (unsigned int step = 0; step < steps; step++) {\n";
            s << "            {\n";
        } else {
            s << "        for (unsigned int step = 0; step < steps; step++) {\n";
            if (E > 1) {
                s << "            #pragma unroll\n";
            }
            s << "            for (unsigned int e = 0; e < " << E << "u; e++) {\n";
            if (E > 1) {
                s << "            if (i + e >= n) continue;\n";
            }
        }
        for (size_t k = 0; k < it.setters.size(); k++) {
            s << "            " << VT << " sv" << k << ";\n";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "            " << VT << " so" << o << ";\n";
        }
    }

//  The body with its window check, the write-back into the lane's state, the end of the pass loop.
    void pass(const bool converge) {
        if (use_shared) {
//  A lane whose denominators leave the window in which the unscaled sequence is the IEEE one
//  (or whose results are not finite) raises a flag the host reports at the next wait():
//  results are then not guaranteed bit-identical and the item should be rebuilt with
//  GFHIP_DIVISION=ieee.  Never observed on the hot-path workloads (|d| spans 1e-30..1e+30).
            s << "            bool bad = false;\n";
            s << "            float dmax = gf_magnitude(" << literal(1.0) << "), dmin = dmax;   // extreme |denominator| of this pass\n";
            s << "            {\n";
            body(true);
            s << "                " << VT << " finite_check = " << literal(0.0) << ";\n";
            for (size_t k = 0; k < it.setters.size(); k++) s << "                finite_check += sv" << k << ";\n";
            for (size_t o = 0; o < it.outputs.size(); o++) s << "                finite_check += so" << o << ";\n";
            s << "                bad = !__builtin_isfinite(" << (packed ? "finite_check.x + finite_check.y" : "finite_check") << ") || !(dmin >= gf_magnitude(" << (f64 ? "0x1p-500" : "0x1p-100f")
              << ")) || !(dmax <= gf_magnitude(" << (f64 ? "0x1p+500" : "0x1p+100f") << "));\n";
            s << "            }\n";
//  Set the status bit once: lanes that find it set only read it (an atomic per flagged lane on
//  one address serialises at ~11 ns each — 0.7 ms for 1e7 flagged lanes).
            s << "            if (__builtin_expect(bad, 0)) {\n"
              << "                if (__hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(flags, 1u);\n"
              << "            }\n";
        } else {
            s << "            {\n";
            body(false);
            s << "            }\n";
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            s << "            " << (packed ? "O" + std::to_string(o) : "o" + std::to_string(o) + "[e]") << " = so" << o << ";\n";
        }
        for (size_t k = 0; k < it.setters.size(); k++) {
            const std::string input = std::to_string(it.setters[k].input);
            s << "            " << (packed ? "V" + input : "v" + input + "[e]") << " = sv" << k << ";\n";
        }
        if (converge) {
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
        } else {
            s << "            }\n";
            s << "        }\n";
            if (packed) {
                for (size_t i = 0; i < it.symbols.size(); i++) {
                    if (out.input_written[i]) s << "        v" << i << "[0] = V" << i << ".x; v" << i << "[1] = V" << i << ".y;\n";
                }
                for (size_t o = 0; o < it.outputs.size(); o++) {
                    s << "        o" << o << "[0] = O" << o << ".x; o" << o << "[1] = O" << o << ".y;\n";
                }
            }
        }
    }

    void stores() {
//  Stores: setters first, then outputs (cpu_context.hpp:522-580).
        std::vector<std::pair<std::string, std::string>> stores;               // (pointer, values)
        for (size_t i = 0; i < it.symbols.size(); i++) {
            if (out.input_written[i]) stores.push_back({"in" + std::to_string(i), "v" + std::to_string(i)});
        }
        for (size_t o = 0; o < it.outputs.size(); o++) {
            stores.push_back({"out" + std::to_string(o), "o" + std::to_string(o)});
        }
        if (E > 1) {
            s << "        if (full) {\n";
            for (auto &st : stores) {
                s << "            {\n"
                  << "                vec_t t;\n"
                  << "                for (unsigned int e = 0; e < " << E << "u; e++) t[e] = " << st.second << "[e];\n"
                  << "                *reinterpret_cast<vec_t *> (" << st.first << " + i) = t;\n"
                  << "            }\n";
            }
            s << "        } else {\n";
            for (auto &st : stores) {
                s << "            for (unsigned int e = 0; e < " << E << "u; e++) if (i + e < n) " << st.first << "[i + e] = " << st.second << "[e];\n";
            }
            s << "        }\n";
        } else if (prefetch && opt.pipeline_tiles) {
//  Keep this tile's results; they are stored from inside the next tile (or after the loop).
            for (size_t i = 0; i < it.symbols.size(); i++) {
                if (out.input_written[i]) s << "        pending_v" << i << " = v" << i << "[0];\n";
            }
            for (size_t o = 0; o < it.outputs.size(); o++) {
                s << "        pending_o" << o << " = o" << o << "[0];\n";
            }
            s << "        pending_index = i;\n        have_pending = true;\n";
            s << "    }\n";
            s << "    if (have_pending) {\n";
            for (size_t i = 0; i < it.symbols.size(); i++) {
                if (out.input_written[i]) s << "        in" << i << "[pending_index] = pending_v" << i << ";\n";
            }
            for (size_t o = 0; o < it.outputs.size(); o++) {
                s << "        out" << o << "[pending_index] = pending_o" << o << ";\n";
            }
            s << "    }\n}\n";
            return;
        } else {
            for (auto &st : stores) {
                s << "        " << st.first << "[i] = " << st.second << "[0];\n";
            }
        }
        s << "    }\n}\n";
        
    }
};

//------------------------------------------------------------------------------
///  @brief Lower one item.
//------------------------------------------------------------------------------
inline lowered lower(const item &original, const codegen_options &opt = codegen_options::from_environment()) {
    item scheduled;
    if (const char *path = std::getenv("GFHIP_ORDER_FILE")) {   // EXPERIMENT: an explicit emission order
        std::ifstream f(path);
        std::vector<uint32_t> order;
        std::vector<uint32_t> fences;                           // 4294967295 in the file = a fence before the next record
        for (uint32_t v; f >> v;) {
            if (v == GFIR_NONE) fences.push_back(static_cast<uint32_t> (order.size())); else order.push_back(v);
        }
        if (order.size() == original.code.size()) {
            scheduled = reorder(original, order);
            scheduled.fences = fences;
        }
    } else if (opt.schedule_for_pressure) {
        scheduled = schedule_for_pressure(original);
    }
    const item &it = scheduled.code.empty() ? original : scheduled;
    lowered out;
    const bool f64 = it.dtype == GFIR_F64;
    const std::string sfx = f64 ? "" : "f";
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

//  Rays per lane.  A lane that owns ONE 4- or 8-byte element issues 4/8-byte loads; small
//  items are HBM bound (xkorc step: 56 B and ~200 flops per particle) and want 16 B per lane
//  per access, so a lane owns 4 (fp32) or 2 (fp64) CONSECUTIVE rays: vector loads/stores and
//  2-4 independent instruction streams per lane.  Large items (the RK4 step) keep one ray per
//  lane — they are register bound.
    uint32_t elements = opt.elements_per_lane;
    if (elements == 0) {
//  Measured (MI355X, xkorc 1e7 particles, loss_kernel 1e6 rays): 2 or 4 rays per lane are
//  not faster than 1 — these items are issue bound by their divisions, not by load width.
        elements = 1;
    }
    if (elements != 1 && elements != 2 && elements != 4) elements = 1;
//  Packed pairs (fp32 only): a lane owns two consecutive rays held as ONE float2, so that the
//  arithmetic of the pass issues as v_pk_add/mul/fma_f32 — two rays per VALU slot instead of
//  one (CDNA's fp32 vector peak is a packed-math figure).  Same IEEE operations per component.
    const bool packed = it.dtype == GFIR_F32 && opt.packed_pairs == 1;
    if (packed) elements = 2;
    out.elements = elements;

    uint32_t park_slots = 0;
    const std::vector<park_plan> plan = plan_parking(it, opt, lds_used, esize, elements, park_slots);
    const size_t park_offset = lds_used;
    lds_used += static_cast<size_t> (park_slots)*opt.block_size*esize;
    out.lds_bytes = lds_used;
    out.park_slots = park_slots;

    std::ostringstream s;
    out.kernel_name = "gfhip_" + it.name;
    emit_prelude(s, it, opt, out.packs.size(), packed);
    const bool use_shared = opt.shared_reciprocal;
//  Two entry points per item: `<name>` runs `steps` passes; `<name>_converge` (items with a
//  setter and an output, one ray per lane) runs the stall loop of workflow.hpp:179-205 PER RAY
//  inside the launch — every lane iterates on its own residual, a wavefront leaves the loop
//  when the ballot of still-active lanes is empty.  That is the reference's converge loop
//  applied to each ray as its own shard; it equals the reference's global-max loop when the
//  rays are identical (the benchmark) and is offered as gfhip_converge_per_ray.
    const bool has_converge = !it.setters.empty() && !it.outputs.empty() && elements == 1 &&
                              it.code.size() <= 1500;
    out.has_converge = has_converge;
    kernel_writer writer{s, it, opt, out, parent, factor, table_pack, table_column, plan, lds_used, park_offset, park_slots,
                         elements, packed, use_shared};
    writer.kernel(false);
    if (has_converge) {
        writer.kernel(true);
    }

    out.source = s.str();
    out.hash = fnv1a(out.source + "|" + compile_flags());
    return out;
}

}  // namespace gfhip

#endif /* gfhip_codegen_hpp */

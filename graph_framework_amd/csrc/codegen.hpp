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
//------------------------------------------------------------------------------
#ifndef gfhip_codegen_hpp
#define gfhip_codegen_hpp

#include <cstdint>
#include <cstdio>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "gfir_item.hpp"

namespace gfhip {

///  All tables of one shape, packed `[cell][column]`.
struct pack {
    uint32_t rows = 1, cols = 1;
    std::vector<uint32_t> tables;       ///< table index per column
    uint32_t stride = 0;                ///< columns padded to an even count (16 B alignment of a cell)
    bool in_lds = false;

    size_t cells() const { return static_cast<size_t> (rows)*cols; }
    size_t elements() const { return cells()*stride; }
};

struct lowered {
    std::string source;
    std::string kernel_name;
    std::vector<pack> packs;
    std::vector<bool> input_written;    ///< input i is the target of a setter
    uint32_t block_size = 256;
    size_t lds_bytes = 0;
    uint64_t hash = 0;
};

struct codegen_options {
    size_t lds_budget = 64*1024;        ///< bytes of LDS the staged packs may use per workgroup
    uint32_t block_size = 256;
};

inline uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

///  Flags the lowering relies on; part of the cache key.
inline const char *compile_flags() {
    return "-O3 -ffp-contract=off --offload-arch=gfx950";
}

//------------------------------------------------------------------------------
///  @brief Lower one item.
//------------------------------------------------------------------------------
inline lowered lower(const item &it, const codegen_options &opt = codegen_options()) {
    lowered out;
    const bool f64 = it.dtype == GFIR_F64;
    const char *real = f64 ? "double" : "float";
    const std::string sfx = f64 ? "" : "f";
    const size_t esize = it.element_size();

    auto literal = [&] (const double v) -> std::string {
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
    };

//  Which inputs are overwritten.
    out.input_written.assign(it.symbols.size(), false);
    for (auto &s : it.setters) {
        out.input_written[s.input] = true;
    }

//  Packs: one per table shape, columns in table order.
    std::map<std::pair<uint32_t, uint32_t>, size_t> pack_of_shape;
    std::vector<uint32_t> table_pack(it.tables.size()), table_column(it.tables.size());
    for (size_t t = 0; t < it.tables.size(); t++) {
        const auto shape = std::make_pair(it.tables[t].rows, it.tables[t].cols);
        auto found = pack_of_shape.find(shape);
        if (found == pack_of_shape.end()) {
            pack p;
            p.rows = shape.first;
            p.cols = shape.second;
            out.packs.push_back(p);
            found = pack_of_shape.insert({shape, out.packs.size() - 1}).first;
        }
        pack &p = out.packs[found->second];
        table_pack[t] = static_cast<uint32_t> (found->second);
        table_column[t] = static_cast<uint32_t> (p.tables.size());
        p.tables.push_back(static_cast<uint32_t> (t));
    }
    size_t lds_used = 0;
    for (auto &p : out.packs) {
        p.stride = static_cast<uint32_t> ((p.tables.size() + 1)/2*2);
    }
//  Stage the smallest packs first while they fit the budget.
    {
        std::vector<size_t> order(out.packs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        for (size_t i = 0; i < order.size(); i++) {
            for (size_t j = i + 1; j < order.size(); j++) {
                if (out.packs[order[j]].elements() < out.packs[order[i]].elements()) std::swap(order[i], order[j]);
            }
        }
        for (size_t i : order) {
            const size_t bytes = out.packs[i].elements()*esize;
            if (lds_used + bytes <= opt.lds_budget) {
                out.packs[i].in_lds = true;
                lds_used += (bytes + 15)/16*16;
            }
        }
    }
    out.lds_bytes = lds_used;
    out.block_size = opt.block_size;

    std::ostringstream s;
    out.kernel_name = "gfhip_" + it.name;
    s << "// Generated by graph_framework_amd (GFIR -> gfx950).  Work item \"" << it.name << "\": "
      << it.code.size() << " nodes, " << it.tables.size() << " tables in " << out.packs.size() << " packs.\n";
    s << "#include <hip/hip_runtime.h>\n";
    s << "typedef " << real << " real;\n";
    s << "extern \"C\" __global__ void __launch_bounds__(" << out.block_size << ")\n" << out.kernel_name << "(";
    for (size_t i = 0; i < it.symbols.size(); i++) {
        s << (out.input_written[i] ? "" : "const ") << "real *__restrict__ in" << i << ", ";
    }
    for (size_t o = 0; o < it.outputs.size(); o++) {
        s << "real *__restrict__ out" << o << ", ";
    }
    for (size_t p = 0; p < out.packs.size(); p++) {
        s << "const real *__restrict__ pack" << p << ", ";
    }
    s << "const unsigned long long n, const unsigned int steps) {\n";

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
    }

    s << "    for (unsigned long long i = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x; i < n;\n"
      << "         i += gridDim.x*static_cast<unsigned long long> (blockDim.x)) {\n";
    for (size_t i = 0; i < it.symbols.size(); i++) {
        std::string symbol = it.symbols[i];
        for (auto &ch : symbol) {
            if (ch == '\\' || ch == '\n') ch = ' ';
        }
        s << "        real v" << i << " = in" << i << "[i];  // " << symbol << "\n";
    }
    for (size_t o = 0; o < it.outputs.size(); o++) {
        s << "        real o" << o << " = 0;\n";
    }
    s << "        for (unsigned int step = 0; step < steps; step++) {\n";

//  Index groups.
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, double, double, double, double> group_key;
    std::map<group_key, std::string> groups;
    size_t group_count = 0;
    auto index_expression = [&] (const uint32_t arg, const double scale, const double offset,
                                 const uint32_t length) -> std::string {
        std::ostringstream e;
        e << "static_cast<unsigned int> (__builtin_fmin" << sfx << "(__builtin_fmax" << sfx << "((r" << arg << " - "
          << literal(offset) << ")/" << literal(scale) << ", " << literal(0.0) << "), "
          << literal(static_cast<double> (length - 1)) << "))";
        return e.str();
    };

    const char *ind = "            ";
    for (size_t i = 0; i < it.code.size(); i++) {
        const gfir_instruction &c = it.code[i];
        switch (c.op) {
            case GFIR_CONST:
                s << ind << "const real r" << i << " = " << literal(c.imm[0]) << ";\n";
                break;
            case GFIR_INPUT:
                s << ind << "const real r" << i << " = v" << c.a << ";\n";
                break;
            case GFIR_ADD:
                s << ind << "const real r" << i << " = r" << c.a << " + r" << c.b << ";\n";
                break;
            case GFIR_SUB:
                s << ind << "const real r" << i << " = r" << c.a << " - r" << c.b << ";\n";
                break;
            case GFIR_MUL:
                s << ind << "const real r" << i << " = r" << c.a << "*r" << c.b << ";\n";
                break;
            case GFIR_DIV:
                s << ind << "const real r" << i << " = r" << c.a << "/r" << c.b << ";\n";
                break;
            case GFIR_FMA:
                s << ind << "const real r" << i << " = __builtin_fma" << sfx << "(r" << c.a << ", r" << c.b
                  << ", r" << c.c << ");\n";
                break;
            case GFIR_SQRT:
                s << ind << "const real r" << i << " = __builtin_sqrt" << sfx << "(r" << c.a << ");\n";
                break;
            case GFIR_POWI: {
                s << ind << "const real r" << i << " = r" << c.a;
                for (uint32_t k = 1; k < c.aux; k++) s << "*r" << c.a;
                s << ";\n";
                break;
            }
            case GFIR_POW:
                s << ind << "const real r" << i << " = pow" << sfx << "(r" << c.a << ", r" << c.b << ");\n";
                break;
            case GFIR_SIN:
                s << ind << "const real r" << i << " = sin" << sfx << "(r" << c.a << ");\n";
                break;
            case GFIR_COS:
                s << ind << "const real r" << i << " = cos" << sfx << "(r" << c.a << ");\n";
                break;
            case GFIR_ATAN2:
                s << ind << "const real r" << i << " = atan2" << sfx << "(r" << c.b << ", r" << c.a << ");\n";
                break;
            case GFIR_EXP:
                s << ind << "const real r" << i << " = exp" << sfx << "(r" << c.a << ");\n";
                break;
            case GFIR_LOG:
                s << ind << "const real r" << i << " = log" << sfx << "(r" << c.a << ");\n";
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
                const uint32_t pi = table_pack[c.aux];
                s << ind << "const real r" << i << " = " << (out.packs[pi].in_lds ? "lds" : "pack") << pi
                  << "[" << g->second << " + " << table_column[c.aux] << "u];\n";
                break;
            }
            default:
                s << ind << "#error unsupported GFIR op\n";
        }
    }
    for (size_t o = 0; o < it.outputs.size(); o++) {
        s << ind << "o" << o << " = r" << it.outputs[o] << ";\n";
    }
    for (auto &st : it.setters) {
        s << ind << "v" << st.input << " = r" << st.value << ";\n";
    }
    s << "        }\n";
//  Stores: setters first, then outputs (cpu_context.hpp:522-580).
    for (size_t i = 0; i < it.symbols.size(); i++) {
        if (out.input_written[i]) {
            s << "        in" << i << "[i] = v" << i << ";\n";
        }
    }
    for (size_t o = 0; o < it.outputs.size(); o++) {
        s << "        out" << o << "[i] = o" << o << ";\n";
    }
    s << "    }\n}\n";

    out.source = s.str();
    out.hash = fnv1a(out.source + "|" + compile_flags());
    return out;
}

}  // namespace gfhip

#endif /* gfhip_codegen_hpp */

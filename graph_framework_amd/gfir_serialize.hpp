//------------------------------------------------------------------------------
///  @file gfir_serialize.hpp
///  @brief Serialize one graph_framework work item (its expression DAG) to GFIR.
///
///  This header is compiled against the reference's own expression-graph
///  headers (graph_framework/node.hpp, arithmetic.hpp, math.hpp,
///  trigonometry.hpp, piecewise.hpp — include them first).  It walks the DAG
///  through the public introspection API only:
///    constant_cast node.hpp:1034, variable_cast :1727, pseudo_variable_cast :1890,
///    add/subtract/multiply/divide_cast arithmetic.hpp:863,1706,2755,3720,
///    fma_cast arithmetic.hpp:5402, sqrt/exp/log/pow_cast math.hpp:321,586,828,1424,
///    sin/cos/atan_cast trigonometry.hpp:262,520,874,
///    piecewise_1D_cast piecewise.hpp:637, piecewise_2D_cast piecewise.hpp:1428.
///  Node identity = pointer (nodes are hash-consed by their factories), visit
///  order = the order leaf_node::compile() recurses, so record i of the GFIR is
///  statement i of the kernel body the reference would have generated.
///
///  Used by hip_context.hpp (in-process lowering) and by the workload exporter.
//------------------------------------------------------------------------------
#ifndef gfir_serialize_h
#define gfir_serialize_h

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include <type_traits>

#include "../include/gfir.h"

namespace gfir {

template<typename T, bool SAFE_MATH=false>
class serializer {
private:
    std::vector<gfir_instruction> code;
    std::map<graph::leaf_node<T, SAFE_MATH> *, uint32_t> slots;
    std::map<graph::leaf_node<T, SAFE_MATH> *, uint32_t> inputs;
    std::vector<std::string> symbols;

    struct table {
        uint32_t rows, cols;
        std::vector<double> data;
    };
    std::vector<table> tables;

    static double real_part(const T value) {
        if constexpr (jit::complex_scalar<T>) {
            return static_cast<double> (std::real(value));
        } else {
            return static_cast<double> (value);
        }
    }
    static double imaginary_part(const T value) {
        if constexpr (jit::complex_scalar<T>) {
            return static_cast<double> (std::imag(value));
        } else {
            return 0.0;
        }
    }

///  The latest draw (GFIR_RANDOM) or the token the first draw hangs on: orders the draws.
    uint32_t last_draw = GFIR_NONE;

    uint32_t emit(const gfir_instruction &i) {
        code.push_back(i);
        return static_cast<uint32_t> (code.size() - 1);
    }

    static gfir_instruction blank(const gfir_op op) {
        gfir_instruction i;
        std::memset(&i, 0, sizeof(i));
        i.op = op;
        i.a = i.b = i.c = GFIR_NONE;
        return i;
    }

//  Tables are de-duplicated by content: the same folded coefficient table is
//  referenced by one gather per RK stage.
    uint32_t add_table(const backend::buffer<T> &b, const uint32_t rows, const uint32_t cols) {
        table t;
        t.rows = rows;
        t.cols = cols;
        if constexpr (jit::complex_scalar<T>) {
            t.data.resize(2*b.size());
            for (size_t i = 0, ie = b.size(); i < ie; i++) {
                t.data[2*i] = real_part(b[i]);
                t.data[2*i + 1] = imaginary_part(b[i]);
            }
        } else {
            t.data.resize(b.size());
            for (size_t i = 0, ie = b.size(); i < ie; i++) {
                t.data[i] = static_cast<double> (b[i]);
            }
        }
        for (size_t i = 0, ie = tables.size(); i < ie; i++) {
            if (tables[i].rows == rows && tables[i].cols == cols &&
                std::memcmp(tables[i].data.data(), t.data.data(), sizeof(double)*t.data.size()) == 0) {
                return static_cast<uint32_t> (i);
            }
        }
        tables.push_back(std::move(t));
        return static_cast<uint32_t> (tables.size() - 1);
    }

    uint32_t lower(graph::shared_leaf<T, SAFE_MATH> n) {
//  random_node::compile registers the TEXT `random(state)` (random.hpp:418): every use of the node
//  is a draw of its own, so it is never looked up.
        if (auto r = graph::random_cast(n); r.get()) {
            return draw();
        }
        auto found = slots.find(n.get());
        if (found != slots.end()) {
            return found->second;
        }
        uint32_t slot = GFIR_NONE;
        if (auto c = graph::constant_cast(n); c.get()) {
            gfir_instruction i = blank(GFIR_CONST);
            const T value = c->evaluate().at(0);
            i.imm[0] = real_part(value);
            i.imm[1] = imaginary_part(value);
            slot = emit(i);
        } else if (auto v = graph::variable_cast(n); v.get()) {
            auto in = inputs.find(n.get());
            if (in == inputs.end()) {
                throw std::runtime_error("gfir: variable is not an input of the work item");
            }
            gfir_instruction i = blank(GFIR_INPUT);
            i.a = in->second;
            slot = emit(i);
        } else if (auto p = graph::pseudo_variable_cast(n); p.get()) {
            slot = lower(p->get_arg());
        } else if (auto x = graph::add_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_ADD);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::subtract_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_SUB);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::multiply_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_MUL);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::divide_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_DIV);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::fma_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_FMA);
            i.a = lower(x->get_left()); i.b = lower(x->get_middle()); i.c = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::sqrt_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_SQRT);
            i.a = lower(x->get_arg());
            slot = emit(i);
        } else if (auto x = graph::pow_cast(n); x.get()) {
            const uint32_t base = lower(x->get_left());
            auto e = graph::constant_cast(x->get_right());
            if (e.get() && e->is_integer()) {
                gfir_instruction i = blank(GFIR_POWI);
                i.a = base;
                i.aux = static_cast<uint32_t> (static_cast<size_t> (std::real(x->get_right()->evaluate().at(0))));
                slot = emit(i);
            } else {
                gfir_instruction i = blank(GFIR_POW);
                i.a = base;
                i.b = lower(x->get_right());
                slot = emit(i);
            }
        } else if (auto x = graph::exp_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_EXP);
            i.a = lower(x->get_arg());
            slot = emit(i);
        } else if (auto x = graph::log_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_LOG);
            i.a = lower(x->get_arg());
            slot = emit(i);
        } else if (auto x = graph::sin_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_SIN);
            i.a = lower(x->get_arg());
            slot = emit(i);
        } else if (auto x = graph::cos_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_COS);
            i.a = lower(x->get_arg());
            slot = emit(i);
        } else if (auto x = graph::atan_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_ATAN2);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            slot = emit(i);
        } else if (auto x = graph::piecewise_1D_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_GATHER1);
            i.a = lower(x->get_arg());
            const backend::buffer<T> data = x->evaluate();
            i.aux = add_table(data, 1, static_cast<uint32_t> (data.size()));
            i.imm[0] = real_part(x->get_scale());
            i.imm[1] = real_part(x->get_offset());
            slot = emit(i);
        } else if (auto x = graph::piecewise_2D_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_GATHER2);
            i.a = lower(x->get_left()); i.b = lower(x->get_right());
            const backend::buffer<T> data = x->evaluate();
            i.aux = add_table(data, static_cast<uint32_t> (x->get_num_rows()),
                              static_cast<uint32_t> (x->get_num_columns()));
            i.imm[0] = real_part(x->get_x_scale());
            i.imm[1] = real_part(x->get_x_offset());
            i.imm[2] = real_part(x->get_y_scale());
            i.imm[3] = real_part(x->get_y_offset());
            slot = emit(i);
        } else if (const uint32_t e = erfi(n); e != none) {
            slot = e;
        } else if (auto x = graph::index_1D_cast(n); x.get()) {
//  v[idx(arg)]: the argument is compiled first, the variable is only named (piecewise.hpp:1530-1575).
            gfir_instruction i = blank(GFIR_INDEX1);
            i.a = lower(x->get_right());
            i.c = input_of(x->get_left());
            i.aux = static_cast<uint32_t> (x->get_size());
            i.imm[0] = real_part(x->get_scale());
            i.imm[1] = real_part(x->get_offset());
            slot = emit(i);
        } else if (auto x = graph::index_2D_cast(n); x.get()) {
            gfir_instruction i = blank(GFIR_INDEX2);
            i.a = lower(x->get_middle());
            i.b = lower(x->get_right());
            i.c = input_of(x->get_left());
            const size_t columns = columns_of_index_2D(n.get());
            i.aux = static_cast<uint32_t> (columns);
            i.reserved = static_cast<uint32_t> (x->get_size()/columns);
            i.imm[0] = real_part(x->get_x_scale());
            i.imm[1] = real_part(x->get_x_offset());
            i.imm[2] = real_part(x->get_y_scale());
            i.imm[3] = real_part(x->get_y_offset());
            slot = emit(i);
        } else {
            throw std::runtime_error("gfir: unsupported node type");
        }
        slots[n.get()] = slot;
        return slot;
    }

//  erfi nodes exist for complex base types only (math.hpp:1439, :1661).
    static constexpr uint32_t none = 0xffffffffu;
    uint32_t erfi(graph::shared_leaf<T, SAFE_MATH> n) {
        if constexpr (jit::complex_scalar<T>) {
            if (auto x = graph::erfi_cast(n); x.get()) {
                gfir_instruction i = blank(GFIR_ERFI);
                i.a = lower(x->get_arg());
                return emit(i);
            }
        }
        return none;
    }

    uint32_t draw() {
        if (last_draw == GFIR_NONE) {
            last_draw = emit(blank(GFIR_CONST));                // the token the first draw hangs on
        }
        gfir_instruction i = blank(GFIR_RANDOM);
        i.a = last_draw;
        last_draw = emit(i);
        return last_draw;
    }

//  What a store of `expression` reads.  Under SAFE_MATH the store is `isnan(e) ? 0 : e`
//  (cpu_context.hpp:530-547) with e the node's register TEXT: for a random node that is two
//  calls of random(state), and the value stored is the second draw.
    uint32_t stored(graph::shared_leaf<T, SAFE_MATH> expression) {
        if constexpr (SAFE_MATH) {
            if (graph::random_cast(expression).get()) {
                draw();
            }
        }
        return lower(expression);
    }

    uint32_t input_of(graph::shared_leaf<T, SAFE_MATH> variable) {
        auto in = inputs.find(variable.get());
        if (in == inputs.end()) {
            throw std::runtime_error("gfir: indexed variable is not an input of the work item");
        }
        return in->second;
    }

//  index_2D_node keeps its column count private (piecewise.hpp:1799, no accessor; even its
//  is_match ignores it): the only public trace is the statement its compile() prints,
//      const T r<node> = v<variable>[<row index>*<columns> + <column index>];
//  (piecewise.hpp:1977-1984), which the caller hands over as `kernel_text`.
    size_t columns_of_index_2D(graph::leaf_node<T, SAFE_MATH> *node) const {
        if (!kernel_text) {
            throw std::runtime_error("gfir: index_2D needs the kernel text to recover its column count");
        }
        const std::string start = " " + jit::to_string('r', node) + " = ";
        const size_t statement = kernel_text->find(start);
        const size_t end = statement == std::string::npos ? statement : kernel_text->find(';', statement);
        if (end != std::string::npos) {
            const std::string text = kernel_text->substr(statement, end - statement);
            const size_t plus = text.find(" + ");
            size_t star = plus == std::string::npos ? plus : text.rfind(")*", plus);
            if (star != std::string::npos) {
                const size_t columns = static_cast<size_t> (std::strtoull(text.c_str() + star + 2, nullptr, 10));
                if (columns) return columns;
            }
        }
        throw std::runtime_error("gfir: cannot find the statement of an index_2D node in the kernel text");
    }

    static void put(std::vector<uint8_t> &out, const void *p, const size_t bytes) {
        const uint8_t *b = static_cast<const uint8_t *> (p);
        out.insert(out.end(), b, b + bytes);
    }

    static void put_string(std::vector<uint8_t> &out, const std::string &s, const bool with_length) {
        const uint32_t padded = static_cast<uint32_t> ((s.size() + 4)/4*4);
        if (with_length) {
            put(out, &padded, 4);
        }
        std::vector<char> buf(padded, '\0');
        std::memcpy(buf.data(), s.data(), s.size());
        put(out, buf.data(), padded);
    }

public:
///  Text the nodes' compile() methods printed for this kernel (only index_2D nodes need it).
    const std::string *kernel_text = nullptr;

//------------------------------------------------------------------------------
///  @brief Serialize a work item.
///
///  Arguments are those of jit::context::add_kernel (jit.hpp:118-126).
///  Setters are lowered before outputs, as add_kernel does (jit.hpp:170-178).
//------------------------------------------------------------------------------
    std::vector<uint8_t> operator()(const std::string &name,
                                    graph::input_nodes<T, SAFE_MATH> in,
                                    graph::output_nodes<T, SAFE_MATH> out,
                                    graph::map_nodes<T, SAFE_MATH> setters) {
        code.clear(); slots.clear(); inputs.clear(); symbols.clear(); tables.clear();
        last_draw = GFIR_NONE;

        for (size_t i = 0, ie = in.size(); i < ie; i++) {
            inputs[in[i].get()] = static_cast<uint32_t> (i);
            symbols.push_back(in[i]->get_symbol());
        }

        std::vector<gfir_setter> set;
        for (auto &[expression, variable] : setters) {
            gfir_setter s;
            s.value = stored(expression);
            s.input = inputs.at(variable.get());
            set.push_back(s);
        }
        std::vector<uint32_t> outs;
        for (auto &o : out) {
            outs.push_back(stored(o));
        }

        std::vector<uint8_t> bytes;
        gfir_header h;
        std::memset(&h, 0, sizeof(h));
        std::memcpy(h.magic, GFIR_MAGIC, 8);
        h.dtype = jit::complex_scalar<T> ? (jit::float_base<T> ? GFIR_C32 : GFIR_C64) : (jit::float_base<T> ? GFIR_F32 : GFIR_F64);
        h.flags = SAFE_MATH ? GFIR_SAFE_MATH : 0;
        h.num_inputs = static_cast<uint32_t> (in.size());
        h.num_outputs = static_cast<uint32_t> (outs.size());
        h.num_setters = static_cast<uint32_t> (set.size());
        h.num_tables = static_cast<uint32_t> (tables.size());
        h.num_instructions = static_cast<uint32_t> (code.size());
        h.name_bytes = static_cast<uint32_t> ((name.size() + 4)/4*4);
        put(bytes, &h, sizeof(h));
        put_string(bytes, name, false);
        for (auto &s : symbols) {
            put_string(bytes, s, true);
        }
        for (auto &t : tables) {
            gfir_table_header th = {t.rows, t.cols};
            put(bytes, &th, sizeof(th));
            put(bytes, t.data.data(), sizeof(double)*t.data.size());
        }
        put(bytes, code.data(), sizeof(gfir_instruction)*code.size());
        put(bytes, outs.data(), sizeof(uint32_t)*outs.size());
        put(bytes, set.data(), sizeof(gfir_setter)*set.size());
        return bytes;
    }
};

}  // namespace gfir

#endif /* gfir_serialize_h */

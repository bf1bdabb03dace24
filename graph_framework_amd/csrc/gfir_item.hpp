//------------------------------------------------------------------------------
///  @file gfir_item.hpp
///  @brief In-memory form of one GFIR work item (include/gfir.h) and its parser.
//------------------------------------------------------------------------------
#ifndef gfir_item_hpp
#define gfir_item_hpp

#include <cctype>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gfir.h"

namespace gfhip {

struct table {
    uint32_t rows, cols;
    std::vector<double> data;               ///< rows*cols values (complex items: (re, im) pairs)
};

struct item {
    uint32_t dtype = GFIR_F64;
    uint32_t flags = 0;                     ///< GFIR_SAFE_MATH
    std::string name;
    std::vector<std::string> symbols;       ///< one per input, kernel argument order
    std::vector<table> tables;
    std::vector<gfir_instruction> code;
    std::vector<uint32_t> outputs;
    std::vector<gfir_setter> setters;

    static size_t element_size(const uint32_t dtype) {
        return dtype == GFIR_F32 ? 4 : dtype == GFIR_C64 ? 16 : 8;
    }
    size_t element_size() const { return element_size(dtype); }
    bool is_complex() const { return dtype == GFIR_C32 || dtype == GFIR_C64; }
    bool base_is_f64() const { return dtype == GFIR_F64 || dtype == GFIR_C64; }
    bool safe_math() const { return flags & GFIR_SAFE_MATH; }
    bool has_random() const {
        for (auto &c : code) {
            if (c.op == GFIR_RANDOM) return true;
        }
        return false;
    }

///  Elements an index node reads of input `i`'s buffer (0 if none does): the buffer bound to
///  that input must hold at least as many.
    size_t indexed_length(const uint32_t i) const {
        size_t length = 0;
        for (auto &c : code) {
            if ((c.op == GFIR_INDEX1 || c.op == GFIR_INDEX2) && c.c == i) {
                const size_t needed = c.op == GFIR_INDEX1 ? c.aux : static_cast<size_t> (c.aux)*c.reserved;
                if (needed > length) length = needed;
            }
        }
        return length;
    }

//------------------------------------------------------------------------------
///  @brief The item as GFIR bytes (include/gfir.h), the inverse of parse().
//------------------------------------------------------------------------------
    std::vector<uint8_t> serialize() const {
        std::vector<uint8_t> out;
        auto put = [&] (const void *src, const size_t n) {
            const uint8_t *b = static_cast<const uint8_t *> (src);
            out.insert(out.end(), b, b + n);
        };
        auto padded = [] (const std::string &text) {
            std::string t = text;
            t.append(4 - t.size()%4, '\0');
            return t;
        };
        const std::string title = padded(name);
        gfir_header h;
        std::memcpy(h.magic, GFIR_MAGIC, 8);
        h.dtype = dtype;
        h.num_inputs = static_cast<uint32_t> (symbols.size());
        h.num_outputs = static_cast<uint32_t> (outputs.size());
        h.num_setters = static_cast<uint32_t> (setters.size());
        h.num_tables = static_cast<uint32_t> (tables.size());
        h.num_instructions = static_cast<uint32_t> (code.size());
        h.name_bytes = static_cast<uint32_t> (title.size());
        h.flags = flags;
        put(&h, sizeof(h));
        put(title.data(), title.size());
        for (auto &symbol : symbols) {
            const std::string text = padded(symbol);
            const uint32_t n = static_cast<uint32_t> (text.size());
            put(&n, 4);
            put(text.data(), n);
        }
        for (auto &t : tables) {
            gfir_table_header th;
            th.rows = t.rows;
            th.cols = t.cols;
            put(&th, sizeof(th));
            put(t.data.data(), sizeof(double)*t.data.size());
        }
        put(code.data(), sizeof(gfir_instruction)*code.size());
        put(outputs.data(), sizeof(uint32_t)*outputs.size());
        put(setters.data(), sizeof(gfir_setter)*setters.size());
        return out;
    }

//------------------------------------------------------------------------------
///  @brief Parse and validate a serialized work item.
///
///  @param[in]  data  Serialized bytes.
///  @param[in]  bytes Number of bytes.
///  @param[out] error Reason on failure.
///  @returns True if the item is well formed.
//------------------------------------------------------------------------------
    bool parse(const void *data, const size_t bytes, std::string &error) {
        const uint8_t *p = static_cast<const uint8_t *> (data);
        size_t pos = 0;
        auto take = [&] (void *dst, const size_t n) -> bool {
            if (n > bytes || pos > bytes - n) {
                error = "GFIR item is truncated";
                return false;
            }
            if (n) std::memcpy(dst, p + pos, n);
            pos += n;
            return true;
        };

        gfir_header h;
        if (!take(&h, sizeof(h))) return false;
        if (std::memcmp(h.magic, GFIR_MAGIC, 8) != 0) {
            error = "not a GFIR item (bad magic)";
            return false;
        }
        if (h.dtype > GFIR_C64) {
            error = "unsupported GFIR dtype";
            return false;
        }
        if (h.flags & ~GFIR_SAFE_MATH) {
            error = "unknown GFIR flags";
            return false;
        }
        dtype = h.dtype;
        flags = h.flags;
        const size_t parts = is_complex() ? 2 : 1;
//  Every count is bounded by the bytes that remain, before anything is allocated.
        if (h.name_bytes > bytes || h.num_inputs > bytes/4 || h.num_tables > bytes/8 ||
            h.num_instructions > bytes/sizeof(gfir_instruction) || h.num_outputs > bytes/4 ||
            h.num_setters > bytes/sizeof(gfir_setter)) {
            error = "GFIR header counts exceed the item size";
            return false;
        }

        std::vector<char> text(h.name_bytes + 1, '\0');
        if (!take(text.data(), h.name_bytes)) return false;
        name = text.data();
//  The name becomes the kernel's identifier in the generated source.
        if (name.empty() || name.size() > 63 || !(std::isalpha(static_cast<unsigned char> (name[0])) || name[0] == '_')) {
            error = "GFIR item name is not an identifier";
            return false;
        }
        for (const char ch : name) {
            if (!(std::isalnum(static_cast<unsigned char> (ch)) || ch == '_')) {
                error = "GFIR item name is not an identifier";
                return false;
            }
        }

        symbols.clear();
        for (uint32_t i = 0; i < h.num_inputs; i++) {
            uint32_t n;
            if (!take(&n, 4)) return false;
            if (n > bytes) {
                error = "GFIR item is truncated";
                return false;
            }
            std::vector<char> s(n + 1, '\0');
            if (!take(s.data(), n)) return false;
            symbols.push_back(s.data());
        }

        tables.assign(h.num_tables, table());
        for (auto &t : tables) {
            gfir_table_header th;
            if (!take(&th, sizeof(th))) return false;
            t.rows = th.rows;
            t.cols = th.cols;
            if (t.rows == 0 || t.cols == 0) {
                error = "empty table in GFIR item";
                return false;
            }
            if (static_cast<size_t> (th.rows)*th.cols > bytes/sizeof(double)/parts) {
                error = "GFIR item is truncated";
                return false;
            }
            t.data.resize(static_cast<size_t> (th.rows)*th.cols*parts);
            if (!take(t.data.data(), sizeof(double)*t.data.size())) return false;
        }

        code.resize(h.num_instructions);
        if (!take(code.data(), sizeof(gfir_instruction)*code.size())) return false;
        outputs.resize(h.num_outputs);
        if (!take(outputs.data(), sizeof(uint32_t)*outputs.size())) return false;
        setters.resize(h.num_setters);
        if (!take(setters.data(), sizeof(gfir_setter)*setters.size())) return false;

//  Validate operand indices: records are in SSA order.
        for (size_t i = 0; i < code.size(); i++) {
            const gfir_instruction &c = code[i];
            auto before = [&] (const uint32_t x) { return x < i; };
            bool ok = true;
            switch (c.op) {
                case GFIR_CONST: break;
                case GFIR_INPUT: ok = c.a < h.num_inputs; break;
                case GFIR_ADD: case GFIR_SUB: case GFIR_MUL: case GFIR_DIV:
                case GFIR_POW: case GFIR_ATAN2:
                    ok = before(c.a) && before(c.b); break;
                case GFIR_FMA: ok = before(c.a) && before(c.b) && before(c.c); break;
                case GFIR_SQRT: case GFIR_SIN: case GFIR_COS: case GFIR_EXP: case GFIR_LOG:
                    ok = before(c.a); break;
                case GFIR_ERFI: ok = before(c.a) && (h.dtype == GFIR_C32 || h.dtype == GFIR_C64); break;
                case GFIR_POWI: ok = before(c.a) && c.aux >= 1 && c.aux <= 64; break;
                case GFIR_GATHER1:
                    ok = before(c.a) && c.aux < h.num_tables && tables[c.aux].rows == 1; break;
                case GFIR_GATHER2:
                    ok = before(c.a) && before(c.b) && c.aux < h.num_tables; break;
                case GFIR_RANDOM: ok = before(c.a); break;
                case GFIR_INDEX1:
                    ok = before(c.a) && c.c < h.num_inputs && c.aux >= 1; break;
                case GFIR_INDEX2:
                    ok = before(c.a) && before(c.b) && c.c < h.num_inputs && c.aux >= 1 && c.reserved >= 1 &&
                         static_cast<uint64_t> (c.aux)*c.reserved <= 0xFFFFFFFFull; break;
                default: ok = false;
            }
            if (!ok) {
                error = "malformed GFIR instruction " + std::to_string(i);
                return false;
            }
        }
        for (auto o : outputs) {
            if (o >= code.size()) {
                error = "GFIR output out of range";
                return false;
            }
        }
        for (auto &s : setters) {
            if (s.value >= code.size() || s.input >= h.num_inputs) {
                error = "GFIR setter out of range";
                return false;
            }
        }
        return true;
    }
};

}  // namespace gfhip

#endif /* gfir_item_hpp */

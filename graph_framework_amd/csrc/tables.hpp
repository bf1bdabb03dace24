//------------------------------------------------------------------------------
///  @file tables.hpp
///  @brief Where the coefficient tables of a work item live: exact compaction (tables that are a
///  constant multiple of another are not stored), AoS packs per table shape, LDS staging.
//------------------------------------------------------------------------------
#ifndef gfhip_tables_hpp
#define gfhip_tables_hpp

#include <cmath>
#include <cstdint>
#include <map>
#include <utility>
#include <vector>

#include "gfir_item.hpp"
#include "options.hpp"

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

struct table_layout {
    std::vector<int> parent;            ///< -1 = stored; else the table it is an exact multiple of
    std::vector<double> factor;
    std::vector<uint32_t> table_pack;   ///< pack of every table ...
    std::vector<uint32_t> table_column; ///< ... and its column there (stored tables)
    std::vector<pack> packs;
    size_t lds_used = 0;                ///< bytes of LDS the staged packs take
};

inline table_layout layout_tables(const item &it, const codegen_options &opt) {
//  Table compaction.  reduce() folds constants into coefficient tables at graph-build time
//  (arithmetic.hpp:192-247), so a kernel gathers many tables that are a constant times
//  another one (45 psi tables, 16 independent).  Where fl(k*parent[c]) == table[c] holds for
//  EVERY cell (checked here, in the item's precision) the table is not stored: its gather
//  becomes k*(gather of the parent) — the same bits, one multiply instead of a load, and the
//  2-D pack of the RK4 kernel shrinks from 360 B to one 128 B line per cell.
    table_layout layout;
    const bool f64 = it.dtype == GFIR_F64;
    const size_t esize = it.element_size();
    std::vector<int> &parent = layout.parent;
    std::vector<double> &factor = layout.factor;
    parent.assign(it.tables.size(), -1);
    factor.assign(it.tables.size(), 1.0);
    if (opt.compact_tables && !it.is_complex()) {
//  A pair of tables costs a full comparison only if it passes a two-cell test first: for to = k*from the cross
//  products to[a]*from[b] and to[b]*from[a] agree (to 1e-5: exactness is judged in the item's precision below;
//  a = the cell of from's largest magnitude, b = half the table further on).  With the thousands of tables of
//  a VMEC ray step the pairs are millions, and nearly all of them end here.
        std::vector<size_t> largest(it.tables.size(), 0);
        for (size_t t = 0; t < it.tables.size(); t++) {
            double best = 0.0;
            for (size_t c = 0; c < it.tables[t].data.size(); c++) {
                if (std::fabs(it.tables[t].data[c]) > best) { best = std::fabs(it.tables[t].data[c]); largest[t] = c; }
            }
        }
        auto may_derive = [&] (const size_t from_index, const size_t to_index) -> bool {
            const table &from = it.tables[from_index], &to = it.tables[to_index];
            if (from.rows != to.rows || from.cols != to.cols) return false;
            const size_t a = largest[from_index], b = (a + from.data.size()/2)%from.data.size();
            if ((from.data[a] == 0.0) != (to.data[a] == 0.0) || (from.data[b] == 0.0) != (to.data[b] == 0.0)) return false;
            const double left = to.data[a]*from.data[b], right = to.data[b]*from.data[a];
            return std::fabs(left - right) <= 1.0E-5*std::fmax(std::fabs(left), std::fabs(right));
        };
        auto derive = [&] (const table &from, const table &to, double &k_out) -> bool {
            if (from.rows != to.rows || from.cols != to.cols) return false;
            size_t arg = 0;
            double best = 0.0;
            for (size_t c = 0; c < from.data.size(); c++) {
                if ((from.data[c] == 0.0) != (to.data[c] == 0.0)) return false;
                if (std::fabs(from.data[c]) > best) { best = std::fabs(from.data[c]); arg = c; }
            }
            if (best == 0.0) return false;
            const double k0 = to.data[arg]/from.data[arg];
            std::vector<double> candidates = {k0, std::nextafter(k0, 1.0E300), std::nextafter(k0, -1.0E300)};
            for (int q = 1; q <= 12; q++) {
                const double p = std::nearbyint(k0*q);
                if (p != 0.0 && std::fabs(p/q - k0) <= 1.0E-12*std::fabs(k0)) candidates.push_back(p/q);
            }
            for (const double k : candidates) {
                bool exact = true;
                for (size_t c = 0; c < from.data.size() && exact; c++) {
                    if (f64) {
                        exact = k*from.data[c] == to.data[c];
                    } else {
                        exact = static_cast<float> (k)*static_cast<float> (from.data[c]) == static_cast<float> (to.data[c]) &&
                                static_cast<double> (static_cast<float> (k)) == k;
                    }
                }
                if (exact) { k_out = k; return true; }
            }
            return false;
        };
//  First pass: a table is derived from an EARLIER table (stored or itself derived; parents
//  always have a smaller index, so there are no cycles).
        for (size_t j = 0; j < it.tables.size(); j++) {
            for (size_t i = 0; i < j; i++) {
                double k;
                if (may_derive(i, j) && derive(it.tables[i], it.tables[j], k)) {
                    parent[j] = static_cast<int> (i);
                    factor[j] = k;
                    break;
                }
            }
        }
//  Second pass: a still-stored table that is an exact multiple of a LATER stored table (e.g.
//  3*c met before c) is re-parented to it; only stored tables become parents here, and they
//  keep no parent of a smaller index, so chains stay acyclic.
        for (size_t j = 0; j < it.tables.size(); j++) {
            if (parent[j] >= 0) continue;
            for (size_t i = j + 1; i < it.tables.size(); i++) {
                if (parent[i] >= 0) continue;
                double k;
                if (may_derive(i, j) && derive(it.tables[i], it.tables[j], k)) {
                    parent[j] = static_cast<int> (i);
                    factor[j] = k;
                    break;
                }
            }
        }
    }
//  Packs: one per table shape, one column per STORED table, in table order.
    std::map<std::pair<uint32_t, uint32_t>, size_t> pack_of_shape;
    std::vector<uint32_t> &table_pack = layout.table_pack, &table_column = layout.table_column;
    table_pack.assign(it.tables.size(), 0);
    table_column.assign(it.tables.size(), 0);
    for (size_t t = 0; t < it.tables.size(); t++) {
        const auto shape = std::make_pair(it.tables[t].rows, it.tables[t].cols);
        auto found = pack_of_shape.find(shape);
        if (found == pack_of_shape.end()) {
            pack p;
            p.rows = shape.first;
            p.cols = shape.second;
            layout.packs.push_back(p);
            found = pack_of_shape.insert({shape, layout.packs.size() - 1}).first;
        }
        pack &p = layout.packs[found->second];
        table_pack[t] = static_cast<uint32_t> (found->second);
        if (parent[t] >= 0) continue;
        table_column[t] = static_cast<uint32_t> (p.tables.size());
        p.tables.push_back(static_cast<uint32_t> (t));
    }
    size_t &lds_used = layout.lds_used;
    for (auto &p : layout.packs) {
        p.stride = static_cast<uint32_t> ((p.tables.size() + 1)/2*2);
    }
//  Stage the smallest packs first while they fit the budget.
    {
        std::vector<size_t> order(layout.packs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        for (size_t i = 0; i < order.size(); i++) {
            for (size_t j = i + 1; j < order.size(); j++) {
                if (layout.packs[order[j]].elements() < layout.packs[order[i]].elements()) std::swap(order[i], order[j]);
            }
        }
        for (size_t i : order) {
            const size_t bytes = layout.packs[i].elements()*esize;
            if (lds_used + bytes <= opt.lds_budget) {
                layout.packs[i].in_lds = true;
                lds_used += (bytes + 15)/16*16;
            }
        }
    }
    return layout;
}

}  // namespace gfhip

#endif /* gfhip_tables_hpp */

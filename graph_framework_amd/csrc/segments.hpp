//------------------------------------------------------------------------------
///  @file segments.hpp
///  @brief Split one work item into consecutive segments that are lowered as kernels of their own.
///
///  Any split of the (topologically ordered) records into consecutive ranges computes the same IEEE
///  values when every value that is defined in one range and used in a later one is handed over
///  through memory.  Two uses:
///
///  * items too large to compile as one kernel.  The reference has no such limit in principle — its
///    kernels are one statement per node whatever the count — but a (cold_plasma x rk4) step on the VMEC
///    equilibrium (equilibrium.hpp:1868-2330; 86 Fourier modes) is 54 k records with 18 k gathers; hipcc
///    needs a time that grows much faster than linearly for one function of that size.  Above
///    `segment_nodes` records an item is cut into pieces of about that size (GFHIP_SEGMENT_NODES);
///  * the experiment VERDICT r2 #2 asks for on the RK4 item: its 16/20/25-value waists (the stage
///    boundaries) give segments that fit 256 registers, i.e. two waves per SIMD, when no IEEE callee is
///    compiled into them (GFHIP_SEGMENTS=<count>).
///
///  A segment is an ordinary item: its symbols are the state inputs it reads plus one input per
///  handed-over value it uses, its outputs are the values it hands over (the last segment: the item's
///  setters and outputs), and it is lowered by the same kernel_writer.  The state arrays are written by
///  the last segment only, so every segment reads the state of the beginning of the pass.  Hand-over
///  buffers are SoA arrays of one chunk of rays; a slot is reused once its value has been read for the
///  last time.  The host walks the ensemble in chunks sized so that the hand-over buffers stay in the
///  256 MB Infinity Cache (gf_hip.cpp).
//------------------------------------------------------------------------------
#ifndef gfhip_segments_hpp
#define gfhip_segments_hpp

#include <algorithm>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "gfir_item.hpp"
#include "options.hpp"
#include "schedule.hpp"

namespace gfhip {

///  One segment as an item, with the meaning of its symbols and outputs.
struct segment {
    item piece;
    std::vector<int> symbol_state;          ///< per symbol of `piece`: index of the original item's input, or -1
    std::vector<int> symbol_slot;           ///< per symbol of `piece`: hand-over slot it reads, or -1
    std::vector<int> symbol_record;         ///< per symbol of `piece`: the record of the split item whose value the slot holds, or -1
    std::vector<int> output_slot;           ///< per output of `piece`: hand-over slot it writes, or -1 (an output of the item)
    std::vector<int> output_original;       ///< per output of `piece`: index of the original item's output, or -1
};

struct segmentation {
    std::vector<segment> segments;
    uint32_t slots = 0;                     ///< hand-over slots (each one element per ray of a chunk)
    std::vector<size_t> cuts;               ///< first record of every segment but the first
    std::vector<size_t> crossing;           ///< values alive across each cut
};

///  Items the split does not handle: complex values, SAFE_MATH guards, random draws, index nodes
///  (their records refer to symbols, not only to records).
inline bool can_split(const item &it) {
    if (it.is_complex() || it.safe_math() || it.has_random()) return false;
    for (auto &c : it.code) {
        if (c.op == GFIR_INDEX1 || c.op == GFIR_INDEX2) return false;
    }
    return !it.code.empty();
}

//------------------------------------------------------------------------------
///  @brief Choose the cuts: `count - 1` positions near the equal-size positions, each where the
///  fewest values are alive across it.
///
///  @param[in] it     Item in emission order.
///  @param[in] count  Number of segments.
///  @param[in] window Half width of the search window around each equal-size position, as a fraction
///                    of the segment length.
//------------------------------------------------------------------------------
inline std::vector<size_t> choose_cuts(const item &it, const size_t count, const double window = 0.45) {
    const size_t n = it.code.size();
    std::vector<size_t> last_use(n, 0);
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = it.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        for (int k = 0; k < operand_count(c.op); k++) last_use[operands[k]] = i;
    }
    for (auto &s : it.setters) last_use[s.value] = n;
    for (auto o : it.outputs) last_use[o] = n;
//  alive[p] = values defined before record p and used at or after it (constants and inputs are free).
    std::vector<int> delta(n + 2, 0);
    for (size_t v = 0; v < n; v++) {
        const uint32_t op = it.code[v].op;
        if (op == GFIR_CONST || op == GFIR_INPUT || last_use[v] <= v) continue;
        delta[v + 1]++;
        delta[std::min(last_use[v], n) + 1]--;
    }
    std::vector<int> alive(n + 1, 0);
    int running = 0;
    for (size_t p = 0; p <= n; p++) {
        running += delta[p];
        alive[p] = running;
    }
    std::vector<size_t> cuts;
    const double length = static_cast<double> (n)/static_cast<double> (count);
    for (size_t k = 1; k < count; k++) {
        const double centre = length*static_cast<double> (k);
        size_t low = static_cast<size_t> (std::max(1.0, centre - window*length));
        size_t high = static_cast<size_t> (std::min(static_cast<double> (n - 1), centre + window*length));
        if (!cuts.empty() && low <= cuts.back()) low = cuts.back() + 1;
        if (high < low) high = low;
        size_t best = low;
        for (size_t p = low; p <= high && p < n; p++) {
            if (alive[p] < alive[best]) best = p;
        }
        cuts.push_back(best);
    }
    return cuts;
}

//------------------------------------------------------------------------------
///  @brief Build the segments of an item for the given cuts.
//------------------------------------------------------------------------------
inline segmentation split_item(const item &it, const std::vector<size_t> &cuts) {
    const size_t n = it.code.size();
    segmentation result;
    result.cuts = cuts;
    std::vector<size_t> bounds = {0};
    for (auto c : cuts) bounds.push_back(c);
    bounds.push_back(n);
    const size_t count = bounds.size() - 1;
    auto segment_of = [&] (const size_t record) -> size_t {
        return static_cast<size_t> (std::upper_bound(bounds.begin(), bounds.end(), record) - bounds.begin()) - 1;
    };

//  Last segment that uses each value (setters and outputs are used by the last segment).
    std::vector<size_t> last_segment(n, 0);
    std::vector<bool> used_later(n, false);
    auto use = [&] (const uint32_t value, const size_t by) {
        const uint32_t op = it.code[value].op;
        if (op == GFIR_CONST || op == GFIR_INPUT) return;
        if (by > segment_of(value)) {
            used_later[value] = true;
            last_segment[value] = std::max(last_segment[value], by);
        }
    };
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = it.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        for (int k = 0; k < operand_count(c.op); k++) use(operands[k], segment_of(i));
    }
    for (auto &s : it.setters) use(s.value, count - 1);
    for (auto o : it.outputs) use(o, count - 1);

//  Hand-over slots: interval colouring over segment numbers (a slot written by segment a and read
//  last by segment b is free for values defined in segments after b).
    std::vector<int> slot_of(n, -1);
    std::vector<size_t> slot_free_after;
    for (size_t v = 0; v < n; v++) {
        if (!used_later[v]) continue;
        const size_t defined = segment_of(v);
        int slot = -1;
        for (size_t k = 0; k < slot_free_after.size(); k++) {
            if (slot_free_after[k] < defined) { slot = static_cast<int> (k); break; }
        }
        if (slot < 0) {
            slot = static_cast<int> (slot_free_after.size());
            slot_free_after.push_back(0);
        }
        slot_free_after[slot] = last_segment[v];
        slot_of[v] = slot;
    }
    result.slots = static_cast<uint32_t> (slot_free_after.size());
    for (auto c : cuts) {
        size_t crossing = 0;
        for (size_t v = 0; v < c; v++) {
            if (used_later[v] && bounds[last_segment[v]] >= c) crossing++;
        }
        result.crossing.push_back(crossing);
    }

    for (size_t k = 0; k < count; k++) {
        segment seg;
        item &piece = seg.piece;
        piece.dtype = it.dtype;
        piece.flags = it.flags;
        piece.name = it.name + "_seg" + std::to_string(k);
        std::map<uint32_t, uint32_t> local;             // original record -> record of the piece
        std::map<uint32_t, uint32_t> table_local;       // original table -> table of the piece
        std::map<int, uint32_t> state_symbol, slot_symbol;
        auto symbol_for_state = [&] (const uint32_t input) -> uint32_t {
            auto found = state_symbol.find(static_cast<int> (input));
            if (found != state_symbol.end()) return found->second;
            const uint32_t index = static_cast<uint32_t> (piece.symbols.size());
            piece.symbols.push_back(it.symbols[input]);
            seg.symbol_state.push_back(static_cast<int> (input));
            seg.symbol_slot.push_back(-1);
            seg.symbol_record.push_back(-1);
            state_symbol[static_cast<int> (input)] = index;
            return index;
        };
        auto symbol_for_slot = [&] (const int slot, const uint32_t record) -> uint32_t {
            auto found = slot_symbol.find(slot);
            if (found != slot_symbol.end()) return found->second;
            const uint32_t index = static_cast<uint32_t> (piece.symbols.size());
            piece.symbols.push_back("handed over, slot " + std::to_string(slot));
            seg.symbol_state.push_back(-1);
            seg.symbol_slot.push_back(slot);
            seg.symbol_record.push_back(static_cast<int> (record));
            slot_symbol[slot] = index;
            return index;
        };
//  The last segment stores the setters: their targets are symbols of the piece whether read or not.
        if (k == count - 1) {
            for (auto &s : it.setters) (void)symbol_for_state(s.input);
        }
//  A record of another segment as an operand: constants and inputs are re-created, anything else
//  is read from its hand-over slot.
        std::function<uint32_t(uint32_t)> resolve = [&] (const uint32_t value) -> uint32_t {
            auto found = local.find(value);
            if (found != local.end()) return found->second;
            const gfir_instruction &c = it.code[value];
            gfir_instruction fresh = c;
            if (c.op == GFIR_INPUT) {
                fresh.a = symbol_for_state(c.a);
            } else if (c.op != GFIR_CONST) {
                fresh = gfir_instruction();
                fresh.op = GFIR_INPUT;
                fresh.a = symbol_for_slot(slot_of[value], value);
                fresh.b = fresh.c = GFIR_NONE;
            }
            piece.code.push_back(fresh);
            local[value] = static_cast<uint32_t> (piece.code.size() - 1);
            return local[value];
        };
        for (size_t i = bounds[k]; i < bounds[k + 1]; i++) {
            gfir_instruction c = it.code[i];
            if (c.op == GFIR_INPUT) {
                c.a = symbol_for_state(c.a);
            } else {
                const int operands = operand_count(c.op);
                if (operands > 0) c.a = resolve(c.a);
                if (operands > 1) c.b = resolve(c.b);
                if (operands > 2) c.c = resolve(c.c);
                if (c.op == GFIR_GATHER1 || c.op == GFIR_GATHER2) {
                    auto found = table_local.find(c.aux);
                    if (found == table_local.end()) {
                        piece.tables.push_back(it.tables[c.aux]);
                        found = table_local.insert({c.aux, static_cast<uint32_t> (piece.tables.size() - 1)}).first;
                    }
                    c.aux = found->second;
                }
            }
            piece.code.push_back(c);
            local[static_cast<uint32_t> (i)] = static_cast<uint32_t> (piece.code.size() - 1);
        }
//  Outputs: what later segments read ...
        for (size_t i = bounds[k]; i < bounds[k + 1]; i++) {
            if (used_later[i]) {
                piece.outputs.push_back(local[static_cast<uint32_t> (i)]);
                seg.output_slot.push_back(slot_of[i]);
                seg.output_original.push_back(-1);
            }
        }
//  ... and, in the last segment, the setters and outputs of the item.
        if (k == count - 1) {
            for (auto &s : it.setters) {
                gfir_setter moved;
                moved.value = resolve(s.value);
                moved.input = symbol_for_state(s.input);
                piece.setters.push_back(moved);
            }
            for (size_t o = 0; o < it.outputs.size(); o++) {
                piece.outputs.push_back(resolve(it.outputs[o]));
                seg.output_slot.push_back(-1);
                seg.output_original.push_back(static_cast<int> (o));
            }
        }
        result.segments.push_back(std::move(seg));
    }
    return result;
}

}  // namespace gfhip

#endif /* gfhip_segments_hpp */

//------------------------------------------------------------------------------
///  @file schedule.hpp
///  @brief Re-order the records of a work item to lower register pressure.
///
///  Any topological order of the DAG computes the same IEEE values.  GFIR arrives
///  in the order leaf_node::compile() recurses (depth first, setters before
///  outputs), which keeps ~174 fp64 values alive at the peak of the RK4 item.
///  A greedy list schedule — always emit the ready node that frees the most
///  operands (an operand is freed when its last consumer is emitted), ties to
///  the node that became ready last (continue the chain just unblocked) —
///  reaches ~150, and the compiler's own scheduler does the rest: the RK4 step
///  drops from 0.242 to 0.223 ms (1e6 rays; AGPR copies 700 -> 540 of 7300
///  VALU instructions).  Tie-breaks tried on the GPU (FIFO, height, eight random
///  seeds, one unit of slack): all within 0.222..0.229 ms, so the deterministic
///  LIFO rule is kept.  The re-ordered item is a plain renumbering; the lowering
///  does not know about it.
//------------------------------------------------------------------------------
#ifndef gfhip_schedule_hpp
#define gfhip_schedule_hpp

#include <set>
#include <vector>

#include "gfir_item.hpp"

namespace gfhip {

inline int operand_count(const uint32_t op) {
    switch (op) {
        case GFIR_CONST: case GFIR_INPUT: return 0;
        case GFIR_FMA: return 3;
        case GFIR_SQRT: case GFIR_POWI: case GFIR_SIN: case GFIR_COS: case GFIR_EXP: case GFIR_LOG: case GFIR_ERFI:
        case GFIR_GATHER1: case GFIR_INDEX1: case GFIR_RANDOM: return 1;
        default: return 2;
    }
}

///  Renumber the records of `in` so that record p of the result is record order[p].
inline item reorder(const item &in, const std::vector<uint32_t> &order) {
    const size_t n = in.code.size();
    std::vector<uint32_t> new_index(n, GFIR_NONE);
    for (size_t p = 0; p < order.size(); p++) new_index[order[p]] = static_cast<uint32_t> (p);
    item out = in;
    for (size_t p = 0; p < order.size(); p++) {
        gfir_instruction c = in.code[order[p]];
        const int count = operand_count(c.op);
        if (count > 0) c.a = new_index[c.a];
        if (count > 1) c.b = new_index[c.b];
        if (count > 2) c.c = new_index[c.c];
        out.code[p] = c;
    }
    for (auto &s : out.setters) s.value = new_index[s.value];
    for (auto &o : out.outputs) o = new_index[o];
    return out;
}

inline item schedule_for_pressure(const item &in) {
    const size_t n = in.code.size();
//  The list schedule below is O(nodes x ready set); items far larger than anything on the path
//  (the RK4 item has 3.9 k nodes) keep their own order rather than stall the lowering.
    if (n > 20000) return in;
    std::vector<std::vector<uint32_t>> users(n);
    std::vector<uint32_t> pending(n, 0), consumers_left(n, 0);
    std::vector<bool> is_root(n, false);
    for (auto &s : in.setters) is_root[s.value] = true;
    for (auto o : in.outputs) is_root[o] = true;
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = in.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        std::set<uint32_t> distinct;
        for (int k = 0; k < operand_count(c.op); k++) distinct.insert(operands[k]);
        pending[i] = static_cast<uint32_t> (distinct.size());
        for (auto o : distinct) {
            users[o].push_back(static_cast<uint32_t> (i));
            consumers_left[o]++;
        }
    }

//  The distinct operands of every record, once (the loop below looks at every ready record at every step).
    std::vector<uint32_t> distinct_operands(3*n, GFIR_NONE);
    std::vector<uint8_t> distinct_count(n, 0);
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = in.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        for (int k = 0; k < operand_count(c.op); k++) {
            bool seen = false;
            for (int j = 0; j < distinct_count[i]; j++) seen = seen || distinct_operands[3*i + j] == operands[k];
            if (!seen) distinct_operands[3*i + distinct_count[i]++] = operands[k];
        }
    }

    std::set<uint32_t> ready;
    std::vector<uint32_t> stamp(n, 0);                   // emission count when the node became ready
    for (size_t i = 0; i < n; i++) {
        if (pending[i] == 0) ready.insert(static_cast<uint32_t> (i));
    }
    std::vector<uint32_t> order;
    order.reserve(n);
    while (!ready.empty()) {
        uint32_t best = *ready.begin();
        int best_score = 1 << 30;
        for (const uint32_t v : ready) {
            int freed = 0;
            for (int k = 0; k < distinct_count[v]; k++) {
                const uint32_t o = distinct_operands[3*static_cast<size_t> (v) + k];
                if (in.code[o].op != GFIR_CONST && consumers_left[o] == 1 && !is_root[o]) freed++;
            }
//  Constants cost nothing; inputs become live only when first read.
            const int grows = (in.code[v].op == GFIR_CONST) ? 0 : 1;
            const int score = grows - freed;
            if (score < best_score || (score == best_score && stamp[v] > stamp[best])) {
                best_score = score;
                best = v;
            }
        }
        ready.erase(best);
        order.push_back(best);
        for (int k = 0; k < distinct_count[best]; k++) consumers_left[distinct_operands[3*static_cast<size_t> (best) + k]]--;
        for (auto u : users[best]) {
            if (--pending[u] == 0) {
                ready.insert(u);
                stamp[u] = static_cast<uint32_t> (order.size());
            }
        }
    }

    return reorder(in, order);
}

///  The same greedy list schedule with its ties (equal score, equal age) broken by a seeded generator.  The peak
///  number of live values varies by a factor of TWO between tie-breaks on the RK4 item (the LDS slots its assembly
///  body needs: 19 to 113 over a hundred seeds), which is what asm_body.hpp's schedule_for_assembly searches.
inline std::vector<uint32_t> list_schedule(const item &in, const uint32_t seed) {
    const size_t n = in.code.size();
    uint64_t state = 0x9E3779B97F4A7C15ull*(seed + 1u);
    auto next = [&state] () {                            // splitmix64: the same sequence on every platform
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30))*0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27))*0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<std::vector<uint32_t>> users(n);
    std::vector<uint32_t> pending(n, 0), consumers_left(n, 0);
    std::vector<bool> is_root(n, false);
    for (auto &s : in.setters) is_root[s.value] = true;
    for (auto o : in.outputs) is_root[o] = true;
    std::vector<uint32_t> distinct_operands(3*n, GFIR_NONE);
    std::vector<uint8_t> distinct_count(n, 0);
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = in.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        for (int k = 0; k < operand_count(c.op); k++) {
            bool seen = false;
            for (int j = 0; j < distinct_count[i]; j++) seen = seen || distinct_operands[3*i + j] == operands[k];
            if (!seen) distinct_operands[3*i + distinct_count[i]++] = operands[k];
        }
        pending[i] = distinct_count[i];
        for (int k = 0; k < distinct_count[i]; k++) {
            users[distinct_operands[3*i + k]].push_back(static_cast<uint32_t> (i));
            consumers_left[distinct_operands[3*i + k]]++;
        }
    }
    std::set<uint32_t> ready;
    std::vector<uint32_t> stamp(n, 0), order, best;
    for (size_t i = 0; i < n; i++) {
        if (pending[i] == 0) ready.insert(static_cast<uint32_t> (i));
    }
    order.reserve(n);
    while (!ready.empty()) {
        best.clear();
        int best_score = 1 << 30;
        uint32_t best_stamp = 0;
        for (const uint32_t v : ready) {
            int freed = 0;
            for (int k = 0; k < distinct_count[v]; k++) {
                const uint32_t o = distinct_operands[3*static_cast<size_t> (v) + k];
                if (in.code[o].op != GFIR_CONST && consumers_left[o] == 1 && !is_root[o]) freed++;
            }
            const int score = (in.code[v].op == GFIR_CONST ? 0 : 1) - freed;
            if (score < best_score || (score == best_score && stamp[v] > best_stamp)) {
                best_score = score;
                best_stamp = stamp[v];
                best.clear();
            }
            if (score == best_score && stamp[v] == best_stamp) best.push_back(v);
        }
        const uint32_t pick = best[static_cast<size_t> (next()%best.size())];
        ready.erase(pick);
        order.push_back(pick);
        for (int k = 0; k < distinct_count[pick]; k++) consumers_left[distinct_operands[3*static_cast<size_t> (pick) + k]]--;
        for (auto u : users[pick]) {
            if (--pending[u] == 0) {
                ready.insert(u);
                stamp[u] = static_cast<uint32_t> (order.size());
            }
        }
    }
    return order;
}

}  // namespace gfhip

#endif /* gfhip_schedule_hpp */

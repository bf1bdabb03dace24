//------------------------------------------------------------------------------
///  @file parking.hpp
///  @brief Which values of a pass wait in a per-lane LDS slot instead of a register.
//------------------------------------------------------------------------------
#ifndef gfhip_parking_hpp
#define gfhip_parking_hpp

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#include "gfir_item.hpp"
#include "options.hpp"
#include "schedule.hpp"

namespace gfhip {

struct park_plan {
    bool parked = false;
    uint32_t slot = 0;
    std::map<size_t, uint32_t> reload_at;       ///< position -> cluster number
};

//  LDS parking.  Measured on MI355X (1e6 rays, ms per RK4 step): none 0.359 (340 B/lane of
//  scratch = 350 MB of HBM writes per step); every value with range > 300 nodes 0.46-0.55
//  (each LDS op costs the single in-order wave an issue slot, like the move it replaces);
//  only values with range > 1500 nodes, <= 32 slots: 0.317 and NO scratch — the default.
//  The RK4 item keeps ~150 fp64 values alive (stage results, the state, shared
//  sub-expressions of the seven partials); at 512 registers per lane the compiler shuttles
//  them through AGPRs (two VALU moves each way) and scratch (HBM write traffic).  Values
//  whose live range is long and whose uses cluster are instead written once to a per-lane
//  LDS slot (`park[slot*block + lane]`, conflict free, LDS pipe instead of VALU) and read
//  back at the first use of every later cluster.  Pure data movement: bits unchanged.
inline std::vector<park_plan> plan_parking(const item &it, const codegen_options &opt, const size_t lds_used,
                                           const size_t esize, uint32_t &park_slots) {
    const size_t node_count = it.code.size();
    std::vector<park_plan> plan(node_count);
    park_slots = 0;
//  A workgroup may declare all 160 KiB of a CU's LDS; stay inside it.
    const size_t lds_capacity = 160*1024;
    const size_t slot_bytes = static_cast<size_t> (opt.block_size)*esize;
    const uint32_t slot_limit = lds_used < lds_capacity
                              ? static_cast<uint32_t> (std::min<size_t> (opt.park_max_slots, (lds_capacity - lds_used)/slot_bytes))
                              : 0;
    if (opt.park_in_lds && slot_limit > 0) {
        std::vector<std::vector<size_t>> uses(node_count);
        auto arity = [] (const uint32_t op) -> int { return operand_count(op); };
        for (size_t i = 0; i < node_count; i++) {
            const gfir_instruction &c = it.code[i];
            const uint32_t operands[3] = {c.a, c.b, c.c};
            for (int k = 0; k < arity(c.op); k++) {
                if (uses[operands[k]].empty() || uses[operands[k]].back() != i) uses[operands[k]].push_back(i);
            }
        }
        for (auto &st : it.setters) uses[st.value].push_back(node_count);
        for (auto o : it.outputs) uses[o].push_back(node_count);

        struct candidate { size_t def, last; uint32_t value; };
        std::vector<candidate> candidates;
        for (size_t v = 0; v < node_count; v++) {
            const uint32_t op = it.code[v].op;
            if (op == GFIR_CONST || op == GFIR_INPUT || uses[v].empty()) continue;
            if (uses[v].back() - v < opt.park_min_range) continue;
            size_t previous = v;
            uint32_t cluster = 0;
            std::map<size_t, uint32_t> reloads;
            const size_t prefetch = opt.park_prefetch < opt.park_window ? opt.park_prefetch : opt.park_window - 1;
            for (const size_t u : uses[v]) {
                if (u - previous > opt.park_window) {
//  Issue the LDS read `prefetch` nodes ahead of the first use of the cluster (there is no
//  other use of the value in that gap: clusters are further apart than the window).
                    reloads[u - prefetch] = ++cluster;
                }
                previous = u;
            }
            if (reloads.empty()) continue;
            plan[v].reload_at = reloads;
            candidates.push_back({v, uses[v].back(), static_cast<uint32_t> (v)});
        }
//  Linear-scan slot allocation in definition order; a slot is free after the last reload.
        std::vector<size_t> slot_free_at;
        for (auto &c : candidates) {
            const size_t last_reload = plan[c.value].reload_at.rbegin()->first;
            uint32_t slot = static_cast<uint32_t> (slot_free_at.size());
            for (uint32_t k = 0; k < slot_free_at.size(); k++) {
                if (slot_free_at[k] < c.def) { slot = k; break; }
            }
            if (slot == slot_free_at.size()) {
                if (slot_free_at.size() >= slot_limit) {
                    plan[c.value].reload_at.clear();
                    continue;
                }
                slot_free_at.push_back(0);
            }
            slot_free_at[slot] = last_reload;
            plan[c.value].parked = true;
            plan[c.value].slot = slot;
        }
        park_slots = static_cast<uint32_t> (slot_free_at.size());
    }
    return plan;
}

//  Parking planned by a replay of the emission order with `capacity` fp64 values in registers (GFHIP_PARK_CAPACITY):
//  a definition that finds the registers full sends the resident value whose next use is farthest away (Belady) to
//  its LDS slot — written once, at its definition: values are immutable — and every later use of a value that is
//  not resident reads it back `park_prefetch` nodes ahead.  Inputs count from the start of the pass, gathers from
//  their first use (codegen.hpp defers the load to it), the shared reciprocals and the cell pointers of the gather
//  groups take room but are never sent away.  The experiment this serves: a pass that fits 256 architectural
//  registers needs no AGPR copies and leaves room for a second wave per SIMD.
inline std::vector<park_plan> plan_parking_belady(const item &it, const codegen_options &opt, const size_t lds_used,
                                                  const size_t esize, uint32_t &park_slots) {
    const size_t n = it.code.size();
    std::vector<park_plan> plan(n);
    park_slots = 0;
    const size_t never = static_cast<size_t> (1) << 60;
    std::vector<std::vector<size_t>> uses(n);
    for (size_t i = 0; i < n; i++) {
        const gfir_instruction &c = it.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        for (int k = 0; k < operand_count(c.op); k++) {
            if (uses[operands[k]].empty() || uses[operands[k]].back() != i) uses[operands[k]].push_back(i);
        }
    }
    for (auto &st : it.setters) uses[st.value].push_back(n);
    for (auto o : it.outputs) uses[o].push_back(n);
//  Room taken at each position by what is never parked: reciprocals (first to last division by a denominator) and
//  gather-group pointers (first to last gather of a group; two groups never share an argument pair and a table shape
//  without sharing the cell, so argument pairs stand in for groups).
    std::vector<int> fixed(n + 1, 0);
    {
        std::map<uint32_t, std::pair<size_t, size_t>> span;
        std::map<std::pair<uint32_t, uint32_t>, std::pair<size_t, size_t>> group_span;
        for (size_t i = 0; i < n; i++) {
            const gfir_instruction &c = it.code[i];
            if (c.op == GFIR_DIV) {
                auto found = span.find(c.b);
                if (found == span.end()) span[c.b] = {i, i}; else found->second.second = i;
            }
            if (c.op == GFIR_GATHER1 || c.op == GFIR_GATHER2) {
                const size_t first_use = uses[i].empty() ? i : uses[i].front();
                const std::pair<uint32_t, uint32_t> key(c.a, c.op == GFIR_GATHER2 ? c.b : GFIR_NONE);
                auto found = group_span.find(key);
                if (found == group_span.end()) group_span[key] = {first_use, first_use};
                else {
                    found->second.first = std::min(found->second.first, first_use);
                    found->second.second = std::max(found->second.second, first_use);
                }
            }
        }
        for (auto &kv : span) for (size_t p = kv.second.first; p <= kv.second.second; p++) fixed[p]++;
        for (auto &kv : group_span) for (size_t p = kv.second.first; p <= kv.second.second; p++) fixed[p]++;
    }
    auto is_value = [&] (const uint32_t v) { return it.code[v].op != GFIR_CONST && !uses[v].empty(); };
    std::vector<size_t> cursor(n, 0);
    auto next_use = [&] (const uint32_t v, const size_t position) {
        const std::vector<size_t> &u = uses[v];
        while (cursor[v] < u.size() && u[cursor[v]] < position) cursor[v]++;
        return cursor[v] < u.size() ? u[cursor[v]] : never;
    };
    std::set<uint32_t> resident;
    std::vector<bool> defined(n, false);
    for (size_t v = 0; v < n; v++) {
        if (it.code[v].op == GFIR_INPUT && is_value(static_cast<uint32_t> (v))) {
            resident.insert(static_cast<uint32_t> (v));
            defined[v] = true;
        }
    }
    const size_t prefetch = opt.park_prefetch;
    std::vector<size_t> last_seen(n, 0);
    size_t reads = 0, writes = 0;
    for (size_t p = 0; p < n; p++) {
        const gfir_instruction &c = it.code[p];
        if (c.op == GFIR_CONST) continue;
        const uint32_t operands[3] = {c.a, c.b, c.c};
        std::set<uint32_t> needed;
        for (int k = 0; k < operand_count(c.op); k++) {
            if (is_value(operands[k])) needed.insert(operands[k]);
        }
        for (const uint32_t o : needed) {
            if (!resident.count(o)) {
                if (defined[o]) {
//  Sent away earlier: read back ahead of this use, but after the previous one.
                    const size_t at = std::max(last_seen[o] + 1, p > prefetch ? p - prefetch : 0);
                    const uint32_t cluster = static_cast<uint32_t> (plan[o].reload_at.size()) + 1;
                    plan[o].reload_at[at] = cluster;
                    reads++;
                } else {
                    defined[o] = true;          // a gather whose load waits for its first use
                }
                resident.insert(o);
            }
            last_seen[o] = p;
        }
        for (const uint32_t o : needed) {
            if (next_use(o, p + 1) == never) resident.erase(o);
        }
        const bool deferred = c.op == GFIR_GATHER1 || c.op == GFIR_GATHER2;
        if (is_value(static_cast<uint32_t> (p)) && c.op != GFIR_INPUT && !deferred) {
            resident.insert(static_cast<uint32_t> (p));
            defined[p] = true;
            last_seen[p] = p;
        }
        while (resident.size() + static_cast<size_t> (fixed[p]) > opt.park_capacity) {
            uint32_t victim = GFIR_NONE;
            size_t farthest = 0;
            for (const uint32_t v : resident) {
                if (v == p || needed.count(v)) continue;
                const size_t u = next_use(v, p + 1);
                if (u >= farthest && u != never) { farthest = u; victim = v; }
            }
            if (victim == GFIR_NONE) break;
            resident.erase(victim);
            if (!plan[victim].parked) {
                plan[victim].parked = true;
                writes++;
            }
        }
    }
//  Slots: linear scan in definition order, a slot is free again after the value's last read.
    const size_t lds_capacity = 160*1024/(opt.waves_per_simd > 1 ? opt.waves_per_simd : 1);
    const size_t slot_bytes = static_cast<size_t> (opt.block_size)*esize;
    const size_t slot_limit = lds_used < lds_capacity ? (lds_capacity - lds_used)/slot_bytes : 0;
    std::vector<size_t> slot_free_at;
    for (size_t v = 0; v < n; v++) {
        if (!plan[v].parked) continue;
        if (plan[v].reload_at.empty()) {            // sent away and never needed again cannot happen; keep the plan consistent
            plan[v].parked = false;
            continue;
        }
        const size_t def = it.code[v].op == GFIR_INPUT ? 0 : v;
        const size_t last_reload = plan[v].reload_at.rbegin()->first;
        uint32_t slot = static_cast<uint32_t> (slot_free_at.size());
        for (uint32_t k = 0; k < slot_free_at.size(); k++) {
            if (slot_free_at[k] < def) { slot = k; break; }
        }
        if (slot == slot_free_at.size()) {
            if (slot_free_at.size() >= slot_limit) {
                plan[v].parked = false;
                plan[v].reload_at.clear();
                continue;
            }
            slot_free_at.push_back(0);
        }
        slot_free_at[slot] = last_reload;
        plan[v].slot = slot;
    }
    park_slots = static_cast<uint32_t> (slot_free_at.size());
    if (std::getenv("GFHIP_PARK_REPORT")) {
        std::fprintf(stderr, "park plan of %s: capacity %u, %zu values parked, %zu reads, %u slots\n", it.name.c_str(),
                     opt.park_capacity, writes, reads, park_slots);
    }
    return plan;
}

}  // namespace gfhip

#endif /* gfhip_parking_hpp */

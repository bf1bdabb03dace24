//------------------------------------------------------------------------------
///  @file parking.hpp
///  @brief Which values of a pass wait in a per-lane LDS slot instead of a register.
//------------------------------------------------------------------------------
#ifndef gfhip_parking_hpp
#define gfhip_parking_hpp

#include <algorithm>
#include <cstdint>
#include <map>
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

}  // namespace gfhip

#endif /* gfhip_parking_hpp */

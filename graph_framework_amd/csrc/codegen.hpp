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
#include "schedule.hpp"

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
    std::vector<int> table_parent;      ///< -1 = stored; else the table it is an exact multiple of
    std::vector<double> table_factor;
    uint32_t block_size = 256;
    size_t lds_bytes = 0;
    uint32_t park_slots = 0;            ///< LDS slots used for parked values
    uint32_t elements = 1;              ///< consecutive rays owned by one lane
    bool has_converge = false;          ///< the module also holds `<name>_converge`
    uint64_t hash = 0;
};

struct codegen_options {
    size_t lds_budget = 64*1024;        ///< bytes of LDS the staged packs may use per workgroup
    uint32_t block_size = 256;
    uint32_t waves_per_simd = 0;        ///< second __launch_bounds__ argument (0 = let the compiler decide)
    bool shared_reciprocal = true;      ///< fp64 divisions by one denominator share its refined reciprocal
    bool pow_three_halves = true;       ///< fp64 pow(x, 1.5) as a compensated x*sqrt(x)
    bool compact_tables = true;         ///< store only tables that are not an exact multiple of another
    bool park_in_lds = true;            ///< very long-lived values wait in LDS instead of AGPRs/scratch (GFHIP_PARK=0 disables)
    uint32_t park_min_range = 1500;     ///< park values whose live range exceeds this many nodes ...
    uint32_t park_window = 100;         ///< ... uses closer than this share one reload
    uint32_t park_max_slots = 32;       ///< LDS slots of block_size elements each
    bool schedule_for_pressure = true;  ///< emit in the pressure-aware order of schedule.hpp (GFHIP_SCHEDULE=source: item order)
    uint32_t elements_per_lane = 0;     ///< rays per lane (0 = auto = 1; 2/4 = vector loads, GFHIP_ELEMENTS_PER_LANE)
    int packed_pairs = -1;              ///< fp32 items: two rays per lane as a float2, arithmetic on v_pk_*_f32
                                        ///< (-1 = auto, GFHIP_PACKED=0/1)
    int division_fixup = -1;            ///< v_div_fixup after each shared-reciprocal quotient: 1 yes, 0 no, -1 auto
    bool prefetch_next_tile = false;    ///< EXPERIMENT: load the next grid-stride tile's inputs before computing this one
    uint32_t prefetch_min_gap = 600;    ///< ... after the latest gather followed by this many gather-free nodes
    bool pipeline_tiles = false;        ///< EXPERIMENT (GFHIP_PIPELINE=1): software-pipelined tiles — the previous tile's
                                        ///< stores and the next tile's loads are issued after the FIRST such gather
    uint32_t sched_barrier_every = 0;   ///< EXPERIMENT: __builtin_amdgcn_sched_barrier(0) every N nodes (0 = none)
    uint32_t park_prefetch = 50;        ///< issue a reload this many nodes before its first use (< window)

//  Environment overrides (they change the generated text, hence the cache key).
    static codegen_options from_environment() {
        codegen_options o;
        if (const char *e = std::getenv("GFHIP_DIVISION")) o.shared_reciprocal = std::string(e) != "ieee";
        if (const char *e = std::getenv("GFHIP_PARK")) o.park_in_lds = std::string(e) != "0";
        if (const char *e = std::getenv("GFHIP_PARK_MIN_RANGE")) o.park_min_range = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_PARK_WINDOW")) o.park_window = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_SCHEDULE")) o.schedule_for_pressure = std::string(e) != "source";
        if (const char *e = std::getenv("GFHIP_ELEMENTS_PER_LANE")) o.elements_per_lane = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_PACKED")) o.packed_pairs = std::atoi(e);
        if (const char *e = std::getenv("GFHIP_DIV_FIXUP")) o.division_fixup = std::string(e) != "0" ? 1 : 0;
        if (const char *e = std::getenv("GFHIP_PREFETCH_NEXT")) o.prefetch_next_tile = std::string(e) == "1";
        if (const char *e = std::getenv("GFHIP_PIPELINE")) {
            o.pipeline_tiles = std::string(e) == "1";
            if (o.pipeline_tiles) o.prefetch_next_tile = true;
        }
        if (const char *e = std::getenv("GFHIP_PREFETCH_MIN_GAP")) o.prefetch_min_gap = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_SCHED_BARRIER")) o.sched_barrier_every = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_PARK_PREFETCH")) o.park_prefetch = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_PARK_MAX_SLOTS")) o.park_max_slots = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_COMPACT_TABLES")) o.compact_tables = std::string(e) != "0";
        if (const char *e = std::getenv("GFHIP_POW")) o.pow_three_halves = std::string(e) != "libm";
        if (const char *e = std::getenv("GFHIP_WAVES_PER_SIMD")) o.waves_per_simd = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_BLOCK_SIZE")) o.block_size = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_LDS_BUDGET")) o.lds_budget = static_cast<size_t> (std::atol(e));
        return o;
    }
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

//  Table compaction.  reduce() folds constants into coefficient tables at graph-build time
//  (arithmetic.hpp:192-247), so a kernel gathers many tables that are a constant times
//  another one (45 psi tables, 16 independent).  Where fl(k*parent[c]) == table[c] holds for
//  EVERY cell (checked here, in the item's precision) the table is not stored: its gather
//  becomes k*(gather of the parent) — the same bits, one multiply instead of a load, and the
//  2-D pack of the RK4 kernel shrinks from 360 B to one 128 B line per cell.
    std::vector<int> parent(it.tables.size(), -1);
    std::vector<double> factor(it.tables.size(), 1.0);
    if (opt.compact_tables) {
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
                if (derive(it.tables[i], it.tables[j], k)) {
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
                if (derive(it.tables[i], it.tables[j], k)) {
                    parent[j] = static_cast<int> (i);
                    factor[j] = k;
                    break;
                }
            }
        }
    }
    out.table_parent = parent;
    out.table_factor = factor;

//  Packs: one per table shape, one column per STORED table, in table order.
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
        if (parent[t] >= 0) continue;
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
    const size_t node_count = it.code.size();
    struct park_plan {
        bool parked = false;
        uint32_t slot = 0;
        std::map<size_t, uint32_t> reload_at;       ///< position -> cluster number
    };
    std::vector<park_plan> plan(node_count);
    uint32_t park_slots = 0;
//  A workgroup may declare all 160 KiB of a CU's LDS; stay inside it.
    const size_t lds_capacity = 160*1024;
    const size_t slot_bytes = static_cast<size_t> (opt.block_size)*esize;
    const uint32_t slot_limit = lds_used < lds_capacity
                              ? static_cast<uint32_t> (std::min<size_t> (opt.park_max_slots, (lds_capacity - lds_used)/slot_bytes))
                              : 0;
    if (opt.park_in_lds && slot_limit > 0 && elements == 1) {
        std::vector<std::vector<size_t>> uses(node_count);
        auto arity = [] (const uint32_t op) -> int {
            switch (op) {
                case GFIR_CONST: case GFIR_INPUT: return 0;
                case GFIR_FMA: return 3;
                case GFIR_SQRT: case GFIR_POWI: case GFIR_SIN: case GFIR_COS: case GFIR_EXP: case GFIR_LOG:
                case GFIR_GATHER1: return 1;
                default: return 2;
            }
        };
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
    const size_t park_offset = lds_used;
    lds_used += static_cast<size_t> (park_slots)*opt.block_size*esize;
    out.lds_bytes = lds_used;
    out.park_slots = park_slots;

    std::ostringstream s;
    out.kernel_name = "gfhip_" + it.name;
    s << "// Generated by graph_framework_amd (GFIR -> gfx950).  Work item \"" << it.name << "\": "
      << it.code.size() << " nodes, " << it.tables.size() << " tables in " << out.packs.size() << " packs.\n";
//  hipRTC predefines the runtime declarations; its include search does not always reach the
//  ROCm headers (a stand-alone process on a box whose /opt/rocm hipRTC has no header path).
    s << "#if !defined(__HIPCC_RTC__)\n#include <hip/hip_runtime.h>\n#endif\n";
    s << "typedef " << real << " real;\n";
    const bool use_shared = opt.shared_reciprocal;
//  v_div_fixup only acts on special operands.  Inside the checked window (finite non-zero
//  denominator, finite results) dropping it changes exactly one thing: a quotient with
//  numerator -0 and a positive denominator comes out +0 instead of -0.  A zero of either sign
//  is the same value to every later operation except a division by it (non-finite, flagged)
//  and atan2, so the fixup (680 of 7700 VALU instructions in the RK4 step, 8 % of its time)
//  is kept only for items that contain an atan2 node.
    bool fixup = opt.division_fixup == 1;
    if (opt.division_fixup < 0) {
        fixup = false;
        for (auto &c : it.code) {
            if (c.op == GFIR_ATAN2) fixup = true;
        }
    }
//  Window check of the denominators: |d| is tracked through an fp32 image that is monotonic in
//  |d| — |d| itself for float, the high dword of a double read as a float (sign, 11 exponent
//  bits, 20 mantissa bits) — so that one v_maximum3_f32 / v_minimum3_f32 (gfx950, abs modifiers
//  free) folds TWO denominators into the running extreme: 1 VALU instruction per denominator
//  instead of 3 with fp64 fmax/fmin.  Both are the IEEE-754-2019 NaN-propagating forms: a
//  high dword that reads as a float NaN (|d| >= 2^1017, infinity, NaN) poisons the accumulator
//  and fails the final comparison, like any other value outside the window.
    s << R"(
__device__ __forceinline__ float gf_magnitude(const float d) { return __builtin_fabsf(d); }
__device__ __forceinline__ float gf_magnitude(const double d) {
    return __builtin_fabsf(__builtin_bit_cast(float, static_cast<unsigned int> (__builtin_bit_cast(unsigned long long, d) >> 32)));
}
)";
    if (!f64) {
        s << (fixup ? "#define GF_FIXUP(q, d, n) __builtin_amdgcn_div_fixupf(q, d, n)\n"
                    : "#define GF_FIXUP(q, d, n) (q)\n");
        s << R"(
// fp32 division as hipcc lowers it (denormals on): scale, r = rcp(d) + one Newton step,
// q = n*r refined by two residual steps, a third residual folded in by div_fmas, un-scale,
// fixup.  Shared per denominator like the fp64 form below; identical bits while no scaling is
// needed: hardware scales when 1/d, n/d or the residual would leave the normal range, i.e.
// |d| outside [2^-126, 2^126], |n/d| outside the normal range, or 0 < |n| < 2^-102.  Checked per
// pass: |d| in [2^-100, 2^100] and finite results; numerators below 2^-102 are not checked
// (their quotients may differ in the last bit).
__device__ __forceinline__ float gf_rcp(const float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float gf_div(const float n, const float d, const float r) {
    const float q0 = n*r;
    const float e0 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-d, q1, n);
    return GF_FIXUP(__builtin_fmaf(e1, r, q1), d, n);
}
)";
    }
    if (f64) {
        s << (fixup ? "#define GF_FIXUP(q, d, n) __builtin_amdgcn_div_fixup(q, d, n)\n"
                    : "#define GF_FIXUP(q, d, n) (q)\n");
        s << R"(
// IEEE fp64 division as hipcc lowers it is: scale, r = rcp(d) refined by two Newton steps,
// q = n*r, e = fma(-d, q, n), q' = fma(e, r, q), un-scale, fix special values.  A work item
// divides many numerators by few denominators (680 divisions, 82 denominators in the RK4
// kernel), so the refinement is done once per denominator.  Without the scaling the sequence
// is the same instruction for instruction, hence bit-identical, while d stays in
// [2^-500, 2^500]; lanes that leave that window are recomputed with the compiler's division.
__device__ __forceinline__ double gf_rcp(const double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double gf_div(const double n, const double d, const double r) {
    const double q = n*r;
    const double e = __builtin_fma(-d, q, n);
    return GF_FIXUP(__builtin_fma(e, r, q), d, n);
}
// pow(x, 1.5) = x*sqrt(x) with the rounding error of the square root carried into the
// product (s + t ~ sqrt(x) to ~100 bits), i.e. rounded once from the exact value almost
// always, as glibc's pow is; ocml's pow is within 1 ulp only and costs ~10x more.
__device__ __forceinline__ double gf_pow_three_halves(const double x) {
    const double s = __builtin_sqrt(x);
    const double t = __builtin_fma(-s, s, x)*(0.5*__builtin_amdgcn_rcp(s));
    const double p = x*s;
    const double c = __builtin_fma(x, s, -p) + x*t;
    return (x > 0.0 && x < __builtin_inf()) ? p + c : pow(x, 1.5);
}
)";
    }
//  Two entry points per item: `<name>` runs `steps` passes; `<name>_converge` (items with a
//  setter and an output, one ray per lane) runs the stall loop of workflow.hpp:179-205 PER RAY
//  inside the launch — every lane iterates on its own residual, a wavefront leaves the loop
//  when the ballot of still-active lanes is empty.  That is the reference's converge loop
//  applied to each ray as its own shard; it equals the reference's global-max loop when the
//  rays are identical (the benchmark) and is offered as gfhip_converge_per_ray.
    const bool has_converge = !it.setters.empty() && !it.outputs.empty() && elements == 1 &&
                              it.code.size() <= 1500;
    if (packed) {
        s << R"(
typedef float real2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ real2 gf_fma2(const real2 a, const real2 b, const real2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ real2 gf_rcp(const real2 d) {
    const real2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const real2 e = gf_fma2(-d, r, (real2)(1.0f));
    return gf_fma2(e, r, r);
}
__device__ __forceinline__ real2 gf_div(const real2 n, const real2 d, const real2 r) {
    const real2 q0 = n*r;
    const real2 e0 = gf_fma2(-d, q0, n);
    const real2 q1 = gf_fma2(e0, r, q0);
    const real2 e1 = gf_fma2(-d, q1, n);
    const real2 q2 = gf_fma2(e1, r, q1);
    return real2{GF_FIXUP(q2.x, d.x, n.x), GF_FIXUP(q2.y, d.y, n.y)};
}
#define GF_PAIR1(name, fn) __device__ __forceinline__ real2 name(const real2 a) { return real2{fn(a.x), fn(a.y)}; }
#define GF_PAIR2(name, fn) __device__ __forceinline__ real2 name(const real2 a, const real2 b) { return real2{fn(a.x, b.x), fn(a.y, b.y)}; }
GF_PAIR1(gf_sqrt2, __builtin_sqrtf)
GF_PAIR1(gf_sin2, sinf)
GF_PAIR1(gf_cos2, cosf)
GF_PAIR1(gf_exp2, expf)
GF_PAIR1(gf_log2, logf)
GF_PAIR2(gf_pow2, powf)
GF_PAIR2(gf_atan22, atan2f)
)";
    }
    out.has_converge = has_converge;
    const std::string VT = packed ? "real2" : "real";         // type of a value of the pass
    auto emit_kernel = [&] (const bool converge) {
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

    const uint32_t E = elements;
    if (E > 1) {
        s << "    typedef real vec_t __attribute__((ext_vector_type(" << E << ")));\n";
        s << "    const bool aligned = ((0";
        for (size_t i = 0; i < it.symbols.size(); i++) s << " | reinterpret_cast<unsigned long long> (in" << i << ")";
        for (size_t o = 0; o < it.outputs.size(); o++) s << " | reinterpret_cast<unsigned long long> (out" << o << ")";
        s << ") & " << (E*esize - 1) << "ull) == 0;\n";
    }
    s << "    const unsigned long long groups = (n + " << (E - 1) << "ull)/" << E << "ull;\n";
    const bool prefetch = opt.prefetch_next_tile && E == 1 && !converge;
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

//  The node-for-node body.  `shared` = divisions through a reciprocal shared by all
//  divisions with the same denominator (see gf_rcp/gf_div in the prelude); otherwise the
//  compiler's IEEE division.
    auto emit_body = [&] (const bool shared) {
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
                name[v] = "r" + std::to_string(v) + "p" + std::to_string(plan[v].reload_at[position]);
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
    };

    if (use_shared) {
//  A lane whose denominators leave the window in which the unscaled sequence is the IEEE one
//  (or whose results are not finite) raises a flag the host reports at the next wait():
//  results are then not guaranteed bit-identical and the item should be rebuilt with
//  GFHIP_DIVISION=ieee.  Never observed on the hot-path workloads (|d| spans 1e-30..1e+30).
        s << "            bool bad = false;\n";
        s << "            float dmax = gf_magnitude(" << literal(1.0) << "), dmin = dmax;   // extreme |denominator| of this pass\n";
        s << "            {\n";
        emit_body(true);
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
        emit_body(false);
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
    };
    emit_kernel(false);
    if (has_converge) {
        emit_kernel(true);
    }

    out.source = s.str();
    out.hash = fnv1a(out.source + "|" + compile_flags());
    return out;
}

}  // namespace gfhip

#endif /* gfhip_codegen_hpp */

//------------------------------------------------------------------------------
///  @file asm_body.hpp
///  @brief The body of a pass as gfx950 assembly with a register assignment of its own.
///
///  hipcc gives the RK4 item all 512 registers of a lane (256 VGPR + 256 AGPR): one wave per SIMD, and —
///  vector instructions cannot name AGPRs — ~700 v_accvgpr copies among the ~6400 vector instructions of a
///  pass.  Every instruction of any kind costs the SIMD one 4-cycle issue slot (profiles/diag/fp64_issue:
///  1.9-2.4 ns per fp64 instruction whether one, two or four waves feed it, dependent or not), so the copies
///  are 10 % of the time, and what a second wave would hide (loads, LDS round trips) another 10 %.
///
///  This writer emits the shared-reciprocal body (codegen.hpp, body(true)) as ONE inline-assembly statement:
///    * the DAG in the pressure order of schedule.hpp, one machine sequence per node — the sequences are the
///      ones hipcc emits for prelude.hpp's gf_rcp / gf_div / gf_sqrt_window / gf_pow_three_halves_window, so
///      every value has the same bits;
///    * values live in a pool of VGPR pairs v[pool_lo:255] (the kernel then fits 256 registers: two waves
///      per SIMD, no AGPRs); a definition that finds the pool full sends the resident value whose next use
///      is farthest away (Belady) to a per-lane LDS slot (ds_write_b64, once: values are immutable) and
///      reads it back ahead of its next use; table values are not written back, they are loaded again;
///    * constants sit in an LRU pool of SGPR pairs (fp64 VOP3 takes no literal on gfx9), one SGPR pair per
///      instruction (constant bus);
///    * loads are issued a few nodes ahead of their first use and waited for with exact vmcnt / lgkmcnt
///      counts; the only hazard the sequences have (a transcendental result read by the next instruction) is
///      covered by an s_nop.
///  The statement reads the state from the compiler's registers ("v" operands), writes the stored values to
///  "=v" operands at its very end and updates the window trackers (dmax, dmin, vmax); the tile loop, the
///  final checks, the redo list and the stores stay HIP (codegen.hpp, role `last`: no IEEE function in the
///  kernel, lanes outside the window are redone by the redo launch).
///
///  Items it does not take (complex, SAFE_MATH, fp32, ops without a sequence here) keep the compiled body.
//------------------------------------------------------------------------------
#ifndef gfhip_asm_body_hpp
#define gfhip_asm_body_hpp

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "gfir_item.hpp"
#include "options.hpp"
#include "schedule.hpp"
#include "tables.hpp"

namespace gfhip {

struct asm_body_text {
    bool ok = false;
    std::string why;                    ///< why the item keeps the compiled body
    std::string statement;              ///< the asm volatile(...) statement
    uint32_t slots = 0;                 ///< LDS slots (block_size elements each) of values sent out of the registers
    size_t vector = 0, scalar = 0, lds_reads = 0, lds_writes = 0, loads = 0, waits = 0;
};

class asm_body_writer {
    const item &it;
    const codegen_options &opt;
    const std::vector<pack> &packs;
    const std::vector<int> &parent;
    const std::vector<double> &factor;
    const std::vector<uint32_t> &table_pack, &table_column;
    const uint32_t block_size;
    const uint32_t slot_limit;

    static constexpr size_t never = static_cast<size_t> (1) << 60;
    static constexpr int sgpr_lo = 56, sgpr_pairs = 22;         ///< s[56:99]: the constants' pool

    const size_t n = it.code.size();
    const uint32_t pool_lo, pairs;

    std::ostringstream a;
    asm_body_text result;

//  Values: [0, n) nodes, [n, 2n) reciprocals of the node n + d, [2n, ...) table values of a cell.
    struct value {
        int reg = -1;                   ///< pair of the pool, -1 = not in a register
        int slot = -1;                  ///< LDS slot, -1 = none
        bool defined = false;
        bool loadable = false;          ///< a table value: loaded (or multiplied from its parent) again instead of parked
        int64_t vm = -1, lgkm = -1;     ///< sequence number of the load in flight into `reg`
        std::string operand;            ///< an input of the statement: its operand name
        std::vector<size_t> uses;
        size_t cursor = 0;
//  table values
        int group = -1;
        uint32_t table = 0;
    };
    std::vector<value> values;
    std::vector<int64_t> alias;                         ///< node -> value it stands for
    std::vector<std::vector<int64_t>> used_at;          ///< position -> values read there
    std::vector<int64_t> owner;                         ///< pair -> value, -1 free, -2 held by the sequence being written
    std::vector<int> free_slots;
    std::set<int64_t> pinned;
    size_t position = 0;

    struct group {
        int offset_pair = -1;           ///< low register: byte offset of the cell in its pack
        uint32_t pack = 0;
        size_t last_use = 0;
        std::map<uint32_t, int64_t> cells;              ///< table -> value
    };
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, double, double, double, double> group_key;
    std::map<group_key, int> group_of_key;
    std::vector<group> groups;
    std::vector<int> node_group;                        ///< gather node -> group

    int64_t vm_issued = 0, lgkm_issued = 0, vm_done = -1, lgkm_done = -1;

//  Constants.
    struct constant_slot { uint64_t bits = 0; bool valid = false; size_t stamp = 0; };
    std::vector<constant_slot> constants = std::vector<constant_slot> (sgpr_pairs);
    size_t constant_clock = 0;

//  Window trackers: operands folded two at a time (v_maximum3_f32 / v_minimum3_f32).
    int pending_track_pair = -1;

  public:
    asm_body_writer(const item &it_, const codegen_options &opt_, const std::vector<pack> &packs_, const std::vector<int> &parent_,
                    const std::vector<double> &factor_, const std::vector<uint32_t> &table_pack_,
                    const std::vector<uint32_t> &table_column_, const uint32_t block_size_, const uint32_t slot_limit_)
        : it(it_), opt(opt_), packs(packs_), parent(parent_), factor(factor_), table_pack(table_pack_), table_column(table_column_),
          block_size(block_size_), slot_limit(slot_limit_), pool_lo(opt_.asm_pool_lo & ~1u), pairs((256 - (opt_.asm_pool_lo & ~1u))/2) {}

    static std::string why_not(const item &it, const codegen_options &opt) {
        if (it.dtype != GFIR_F64) return "not an fp64 item";
        if (it.safe_math() || it.has_random()) return "SAFE_MATH or random draws";
        if (opt.division != division_mode::shared) return "division mode is not `shared`";
        if (opt.division_fixup == 1) return "v_div_fixup requested";
        for (size_t i = 0; i < it.code.size(); i++) {
            const gfir_instruction &c = it.code[i];
            switch (c.op) {
                case GFIR_CONST: case GFIR_INPUT: case GFIR_ADD: case GFIR_SUB: case GFIR_MUL: case GFIR_FMA: case GFIR_GATHER1: case GFIR_GATHER2:
                    break;
                case GFIR_POWI:
                    if (c.aux < 2) return "pow with an integer exponent below 2";
                    if (it.code[c.a].op == GFIR_CONST) return "power of a constant";
                    break;
                case GFIR_DIV:
                    if (it.code[c.b].op == GFIR_CONST) return "division by a constant";
                    break;
                case GFIR_SQRT:
                    if (it.code[c.a].op == GFIR_CONST) return "square root of a constant";
                    break;
                case GFIR_POW:
                    if (!(opt.pow_three_halves && it.code[c.b].op == GFIR_CONST && it.code[c.b].imm[0] == 1.5)) return "pow with an exponent other than 1.5";
                    if (it.code[c.a].op == GFIR_CONST) return "power of a constant";
                    break;
                default:
                    return "an operation without a sequence in asm_body.hpp";
            }
        }
        for (size_t i = 0; i < it.symbols.size(); i++) {
            if (it.indexed_length(static_cast<uint32_t> (i))) return "indexed inputs";
        }
        return "";
    }

  private:
//------------------------------------------------------------------------------
//  Text.
//------------------------------------------------------------------------------
//  A line of the statement; `note` (what the instruction defines, sends to LDS or brings back) rides along as an
//  assembler comment: tests/asm_symbolic.py replays the statement on symbolic values and holds every definition against
//  the item's DAG.
    std::string note;
    void annotate(const std::string &text) { note = text; }
    void line(const std::string &text) {
        a << "                \"" << text;
        if (!note.empty()) a << " ; " << note;
        note.clear();
        a << "\\n\"\n";
    }
    std::string value_name(const int64_t id) const {
        if (id < static_cast<int64_t> (n)) return "r" + std::to_string(id);
        if (id < static_cast<int64_t> (2*n)) return "q" + std::to_string(id - static_cast<int64_t> (n));
        const value &v = values[static_cast<size_t> (id)];
        return "c" + std::to_string(v.group) + "_" + std::to_string(v.table);
    }
    void vector_op(const std::string &text) { line(text); result.vector++; }
    void scalar_op(const std::string &text) { line(text); result.scalar++; }
    static std::string pair_name(const int p, const uint32_t lo) {
        const uint32_t r = lo + 2u*static_cast<uint32_t> (p);
        return "v[" + std::to_string(r) + ":" + std::to_string(r + 1) + "]";
    }
    std::string pair_name(const int p) const { return pair_name(p, pool_lo); }
    std::string low_name(const int p) const { return "v" + std::to_string(pool_lo + 2u*static_cast<uint32_t> (p)); }
    std::string high_name(const int p) const { return "v" + std::to_string(pool_lo + 2u*static_cast<uint32_t> (p) + 1u); }
    static std::string hex32(const uint32_t v) {
        char buffer[16];
        std::snprintf(buffer, sizeof(buffer), "0x%x", v);
        return buffer;
    }

//------------------------------------------------------------------------------
//  Uses.
//------------------------------------------------------------------------------
    size_t next_use(const int64_t id, const size_t from) {
        value &v = values[static_cast<size_t> (id)];
        while (v.cursor < v.uses.size() && v.uses[v.cursor] < from) v.cursor++;
        return v.cursor < v.uses.size() ? v.uses[v.cursor] : never;
    }
    void add_use(const int64_t id, const size_t at) {
        std::vector<size_t> &u = values[static_cast<size_t> (id)].uses;
        if (u.empty() || u.back() != at) u.push_back(at);
        if (at < n) {
            std::vector<int64_t> &here = used_at[at];
            if (std::find(here.begin(), here.end(), id) == here.end()) here.push_back(id);
        }
    }

    int64_t cell_value(const int g, const uint32_t t) {
        auto found = groups[static_cast<size_t> (g)].cells.find(t);
        if (found != groups[static_cast<size_t> (g)].cells.end()) return found->second;
        value v;
        v.loadable = true;
        v.group = g;
        v.table = t;
        values.push_back(v);
        const int64_t id = static_cast<int64_t> (values.size()) - 1;
        groups[static_cast<size_t> (g)].cells[t] = id;
        return id;
    }

//  Who reads what where.  A gather node stands for the table value of its group's cell; a derived table's value
//  reads its parent's at each of its own uses (it may have to be made again after it lost its register).
    void analyse() {
        values.assign(2*n, value());
        alias.assign(n, -1);
        used_at.assign(n, std::vector<int64_t> ());
        node_group.assign(n, -1);
        for (size_t i = 0; i < n; i++) {
            const gfir_instruction &c = it.code[i];
            alias[i] = static_cast<int64_t> (i);
            if (c.op == GFIR_GATHER1 || c.op == GFIR_GATHER2) {
                const table &t = it.tables[c.aux];
                const bool two = c.op == GFIR_GATHER2;
                const group_key key(c.a, two ? c.b : GFIR_NONE, t.rows, t.cols, c.imm[0], c.imm[1], two ? c.imm[2] : 0.0, two ? c.imm[3] : 0.0);
                auto found = group_of_key.find(key);
                if (found == group_of_key.end()) {
                    group g;
                    g.pack = table_pack[c.aux];
                    groups.push_back(g);
                    found = group_of_key.insert({key, static_cast<int> (groups.size()) - 1}).first;
                }
                node_group[i] = found->second;
                alias[i] = cell_value(found->second, c.aux);
            }
        }
        std::vector<bool> group_seen(groups.size(), false);
        for (size_t i = 0; i < n; i++) {
            const gfir_instruction &c = it.code[i];
            if (c.op == GFIR_CONST || c.op == GFIR_INPUT) continue;
            if (c.op == GFIR_GATHER1 || c.op == GFIR_GATHER2) {
//  The arguments are read where the group's cell is found: at its first gather.
                const size_t g = static_cast<size_t> (node_group[i]);
                if (!group_seen[g]) {
                    group_seen[g] = true;
                    if (it.code[c.a].op != GFIR_CONST) add_use(alias[c.a], i);
                    if (c.op == GFIR_GATHER2 && it.code[c.b].op != GFIR_CONST) add_use(alias[c.b], i);
                }
                continue;
            }
            const uint32_t operands[3] = {c.a, c.b, c.c};
            for (int k = 0; k < operand_count(c.op); k++) {
                if (it.code[operands[k]].op != GFIR_CONST) add_use(alias[operands[k]], i);
            }
            if (c.op == GFIR_DIV) add_use(static_cast<int64_t> (n + c.b), i);
        }
        for (auto &st : it.setters) {
            if (it.code[st.value].op != GFIR_CONST) add_use(alias[st.value], n);
        }
        for (auto o : it.outputs) {
            if (it.code[o].op != GFIR_CONST) add_use(alias[o], n);
        }
//  Parents of derived table values, from the leaves up (a value made later appends its parent's uses).
        for (size_t k = 2*n; k < values.size(); k++) {
            int64_t child = static_cast<int64_t> (k);
            while (parent[values[static_cast<size_t> (child)].table] >= 0) {
                const int g = values[static_cast<size_t> (child)].group;
                const int64_t up = cell_value(g, static_cast<uint32_t> (parent[values[static_cast<size_t> (child)].table]));
                std::vector<size_t> merged = values[static_cast<size_t> (up)].uses;
                for (const size_t u : values[static_cast<size_t> (k)].uses) merged.push_back(u);
                std::sort(merged.begin(), merged.end());
                merged.erase(std::unique(merged.begin(), merged.end()), merged.end());
                values[static_cast<size_t> (up)].uses = merged;
                for (const size_t u : values[static_cast<size_t> (k)].uses) {
                    if (u < n && std::find(used_at[u].begin(), used_at[u].end(), up) == used_at[u].end()) used_at[u].push_back(up);
                }
                child = up;
            }
        }
        for (auto &g : groups) {
            for (auto &kv : g.cells) {
                for (const size_t u : values[static_cast<size_t> (kv.second)].uses) g.last_use = std::max(g.last_use, u);
            }
        }
    }

//------------------------------------------------------------------------------
//  Registers.
//------------------------------------------------------------------------------
    bool in_flight(const value &v) const { return v.vm > vm_done || v.lgkm > lgkm_done; }

    void flush_track() {
        if (pending_track_pair < 0) return;
        const std::string h = "abs(" + high_name(pending_track_pair) + ")";
        vector_op("v_maximum3_f32 %[dmax], %[dmax], " + h + ", " + h);
        vector_op("v_minimum3_f32 %[dmin], %[dmin], " + h + ", " + h);
        pending_track_pair = -1;
    }
//  |x| of a denominator or of a square root's argument joins the window check.
    void track(const int pair) {
        if (pending_track_pair < 0) {
            pending_track_pair = pair;
            return;
        }
        const std::string first = "abs(" + high_name(pending_track_pair) + ")", second = "abs(" + high_name(pair) + ")";
        vector_op("v_maximum3_f32 %[dmax], %[dmax], " + first + ", " + second);
        vector_op("v_minimum3_f32 %[dmin], %[dmin], " + first + ", " + second);
        pending_track_pair = -1;
    }

    void release_pair(const int p) {
        if (p == pending_track_pair) flush_track();
        owner[static_cast<size_t> (p)] = -1;
    }

    std::string slot_address(const int slot, std::string &base) const {
//  (the offset field of an LDS instruction has 16 bits: one base register per 64 KB of slots)
        const uint32_t per_base = 65536u/(block_size*8u);
        base = "%[park" + std::to_string(static_cast<uint32_t> (slot)/per_base) + "]";
        return std::to_string((static_cast<uint32_t> (slot)%per_base)*block_size*8u);
    }

//  A pair for a new value.  `horizon`: when asked ahead of time (a load issued early), only values whose next use
//  lies beyond it may lose their register.  Returns -1 when nothing can be had.
    int take_pair(const size_t horizon = 0) {
        for (size_t p = 0; p < owner.size(); p++) {
            if (owner[p] == -1) {
                owner[p] = -2;
                return static_cast<int> (p);
            }
        }
        int best = -1;
        size_t farthest = 0;
        bool best_needs_slot = true;
        for (size_t p = 0; p < owner.size(); p++) {
            const int64_t id = owner[p];
            if (id < 0 || pinned.count(id)) continue;
            value &v = values[static_cast<size_t> (id)];
            if (in_flight(v)) continue;
            const size_t u = next_use(id, position);
            if (horizon && u <= horizon) continue;
            const bool needs_slot = !v.loadable && v.slot < 0 && v.operand.empty();
            if (needs_slot && free_slots.empty() && result.slots >= slot_limit) continue;
            if (best < 0 || u > farthest || (u == farthest && best_needs_slot && !needs_slot)) {
                best = static_cast<int> (p);
                farthest = u;
                best_needs_slot = needs_slot;
            }
        }
        if (best < 0) return -1;
        const int64_t id = owner[static_cast<size_t> (best)];
        value &v = values[static_cast<size_t> (id)];
        if (best == pending_track_pair) flush_track();
        if (!v.loadable && v.slot < 0) {
            if (free_slots.empty()) {
                v.slot = static_cast<int> (result.slots++);
            } else {
                v.slot = free_slots.back();
                free_slots.pop_back();
            }
            std::string base;
            const std::string offset = slot_address(v.slot, base);
            annotate("spill " + value_name(id));
            line("ds_write_b64 " + base + ", " + pair_name(best) + " offset:" + offset);
            lgkm_issued++;
            result.lds_writes++;
        }
        v.reg = -1;
        owner[static_cast<size_t> (best)] = -2;
        return best;
    }

    int need_pair() {
        const int p = take_pair();
        if (p < 0 && result.why.empty()) result.why = "the register pool and the LDS slots are both full";
        return p < 0 ? 0 : p;
    }

    void wait_for(value &v) {
        if (v.vm > vm_done) {
            const int64_t allowed = std::min<int64_t> (vm_issued - 1 - v.vm, 63);
            line("s_waitcnt vmcnt(" + std::to_string(allowed) + ")");
            vm_done = vm_issued - 1 - allowed;
            result.waits++;
        }
        if (v.lgkm > lgkm_done) {
            const int64_t allowed = std::min<int64_t> (lgkm_issued - 1 - v.lgkm, 15);
            line("s_waitcnt lgkmcnt(" + std::to_string(allowed) + ")");
            lgkm_done = lgkm_issued - 1 - allowed;
            result.waits++;
        }
    }

//  Start bringing a value into a register (a load that may still be in flight when this returns).
    bool fetch(const int64_t id, const size_t horizon = 0) {
        value &v = values[static_cast<size_t> (id)];
        if (v.reg >= 0 || !v.operand.empty()) return true;
        if (v.loadable) {
            const group &g = groups[static_cast<size_t> (v.group)];
            if (g.offset_pair < 0) return false;                // the cell is not known yet
            if (parent[v.table] >= 0) {
                if (horizon) return false;                       // made where it is used
                const int64_t up = cell_value(v.group, static_cast<uint32_t> (parent[v.table]));
                const bool was_pinned = pinned.count(up) != 0;
                pinned.insert(up);
                fetch(up);
                wait_for(values[static_cast<size_t> (up)]);
                const int p = need_pair();
                const std::string scaled = constant_operand(factor[v.table], nullptr);
                annotate("def " + value_name(id) + " = " + value_name(up) + " * " + std::to_string(bits_of(factor[v.table])));
                vector_op("v_mul_f64 " + pair_name(p) + ", " + name_of(up) + ", " + scaled);
                if (!was_pinned) pinned.erase(up);
                value &again = values[static_cast<size_t> (id)];
                again.reg = p;
                again.defined = true;
                owner[static_cast<size_t> (p)] = id;
                return true;
            }
            const pack &pk = packs[g.pack];
//  Two neighbouring columns of a global pack that are both wanted soon come with ONE 16-byte load into two adjacent
//  free pairs (half the load instructions of a cell; with incoherent rays every load of a wave touches 64 cache lines).
            if (!pk.in_lds && opt.asm_wide_loads) {
                const uint32_t column = table_column[v.table];
                int64_t partner = -1;
                for (auto &kv : g.cells) {
                    const value &w = values[static_cast<size_t> (kv.second)];
                    if (kv.second == id || parent[w.table] >= 0 || w.reg >= 0) continue;
                    const uint32_t other = table_column[w.table];
                    if (other != column + 1 && other + 1 != column) continue;
                    const size_t u = next_use(kv.second, position);
                    if (u == never || u > position + 2*static_cast<size_t> (opt.asm_load_ahead) + 8) continue;
                    partner = kv.second;
                    break;
                }
                int first = -1;
                if (partner >= 0) {
                    for (size_t q = 0; q + 1 < owner.size(); q++) {
                        if (owner[q] == -1 && owner[q + 1] == -1) { first = static_cast<int> (q); break; }
                    }
                }
                if (first >= 0) {
                    const bool id_is_low = table_column[values[static_cast<size_t> (partner)].table] > column;
                    const int64_t low_id = id_is_low ? id : partner, high_id = id_is_low ? partner : id;
                    const uint32_t low_column = table_column[values[static_cast<size_t> (low_id)].table];
                    const uint32_t r = pool_lo + 2u*static_cast<uint32_t> (first);
                    annotate("def " + value_name(low_id) + " " + value_name(high_id));
                    line("global_load_dwordx4 v[" + std::to_string(r) + ":" + std::to_string(r + 3) + "], " + low_name(g.offset_pair) + ", %[pack" +
                         std::to_string(g.pack) + "] offset:" + std::to_string(low_column*8u));
                    const int64_t sequence = vm_issued++;
                    result.loads++;
                    const int64_t both[2] = {low_id, high_id};
                    for (int k = 0; k < 2; k++) {
                        value &loaded = values[static_cast<size_t> (both[k])];
                        loaded.vm = sequence;
                        loaded.reg = first + k;
                        loaded.defined = true;
                        owner[static_cast<size_t> (first + k)] = both[k];
                    }
                    return true;
                }
            }
            const int p = horizon ? take_pair(horizon) : need_pair();
            if (p < 0) return false;
            const std::string offset = std::to_string(table_column[v.table]*8u);
            annotate("def " + value_name(id));
            if (pk.in_lds) {
                line("ds_read_b64 " + pair_name(p) + ", " + low_name(g.offset_pair) + " offset:" + offset);
                values[static_cast<size_t> (id)].lgkm = lgkm_issued++;
                result.lds_reads++;
            } else {
                line("global_load_dwordx2 " + pair_name(p) + ", " + low_name(g.offset_pair) + ", %[pack" + std::to_string(g.pack) + "] offset:" + offset);
                values[static_cast<size_t> (id)].vm = vm_issued++;
                result.loads++;
            }
            value &again = values[static_cast<size_t> (id)];
            again.reg = p;
            again.defined = true;
            owner[static_cast<size_t> (p)] = id;
            return true;
        }
        if (!v.defined || v.slot < 0) return false;
        const int p = horizon ? take_pair(horizon) : need_pair();
        if (p < 0) return false;
        std::string base;
        const std::string offset = slot_address(values[static_cast<size_t> (id)].slot, base);
        annotate("fill " + value_name(id));
        line("ds_read_b64 " + pair_name(p) + ", " + base + " offset:" + offset);
        value &again = values[static_cast<size_t> (id)];
        again.lgkm = lgkm_issued++;
        again.reg = p;
        owner[static_cast<size_t> (p)] = id;
        result.lds_reads++;
//  The slot is free again (LDS operations of a wave complete in order): a value that loses its register a second
//  time is written a second time — slots are the scarcer resource.
        free_slots.push_back(again.slot);
        again.slot = -1;
        return true;
    }

    std::string name_of(const int64_t id) const {
        const value &v = values[static_cast<size_t> (id)];
        return v.operand.empty() ? pair_name(v.reg) : v.operand;
    }

//  Loads ahead of the nodes that read them.
    void look_ahead() {
        const size_t far = std::min(n, position + 1 + opt.asm_load_ahead), near = std::min(n, position + 1 + opt.asm_reload_ahead);
        for (size_t j = position + 1; j < far; j++) {
            for (const int64_t id : used_at[j]) {
                const value &v = values[static_cast<size_t> (id)];
                if (v.reg >= 0 || !v.operand.empty()) continue;
                if (v.loadable) {
                    if (parent[v.table] < 0 && !packs[groups[static_cast<size_t> (v.group)].pack].in_lds) fetch(id, j + opt.asm_load_ahead);
                    else if (parent[v.table] < 0 && j < near) fetch(id, j + opt.asm_load_ahead);
                } else if (j < near && v.defined && v.slot >= 0) {
                    fetch(id, j + opt.asm_load_ahead);
                }
            }
        }
    }

//------------------------------------------------------------------------------
//  Constants.
//------------------------------------------------------------------------------
    static uint64_t bits_of(const double v) {
        uint64_t b;
        std::memcpy(&b, &v, sizeof(b));
        return b;
    }
    static const char *inline_constant(const uint64_t bits) {
        switch (bits) {
            case 0x0000000000000000ull: return "0";
            case 0x3fe0000000000000ull: return "0.5";
            case 0xbfe0000000000000ull: return "-0.5";
            case 0x3ff0000000000000ull: return "1.0";
            case 0xbff0000000000000ull: return "-1.0";
            case 0x4000000000000000ull: return "2.0";
            case 0xc000000000000000ull: return "-2.0";
            case 0x4010000000000000ull: return "4.0";
            case 0xc010000000000000ull: return "-4.0";
            default: return nullptr;
        }
    }
//  The operand text of a constant: an inline constant or an SGPR pair of the pool (loaded here if it is not there).
//  `taken`: the pool entry another operand of the same instruction already uses — an instruction reads ONE SGPR
//  pair (constant bus), a second constant goes through a register pair (`spare`).
    std::string constant_operand(const double v, int *taken, std::vector<int> *spare = nullptr) {
        const uint64_t bits = bits_of(v);
        if (const char *text = inline_constant(bits)) return text;
        int found = -1, oldest = -1;
        for (int k = 0; k < sgpr_pairs; k++) {
            if (constants[static_cast<size_t> (k)].valid && constants[static_cast<size_t> (k)].bits == bits) found = k;
        }
        if (found < 0) {
            for (int k = 0; k < sgpr_pairs; k++) {
                if (taken && *taken == k) continue;
                if (!constants[static_cast<size_t> (k)].valid) { oldest = k; break; }
                if (oldest < 0 || constants[static_cast<size_t> (k)].stamp < constants[static_cast<size_t> (oldest)].stamp) oldest = k;
            }
            found = oldest;
            constants[static_cast<size_t> (found)].bits = bits;
            constants[static_cast<size_t> (found)].valid = true;
            scalar_op("s_mov_b32 s" + std::to_string(sgpr_lo + 2*found) + ", " + hex32(static_cast<uint32_t> (bits)));
            scalar_op("s_mov_b32 s" + std::to_string(sgpr_lo + 2*found + 1) + ", " + hex32(static_cast<uint32_t> (bits >> 32)));
        }
        constants[static_cast<size_t> (found)].stamp = ++constant_clock;
        const std::string name = "s[" + std::to_string(sgpr_lo + 2*found) + ":" + std::to_string(sgpr_lo + 2*found + 1) + "]";
        if (taken && *taken >= 0 && *taken != found) {
            const int p = need_pair();
            vector_op("v_mov_b32_e32 " + low_name(p) + ", s" + std::to_string(sgpr_lo + 2*found));
            vector_op("v_mov_b32_e32 " + high_name(p) + ", s" + std::to_string(sgpr_lo + 2*found + 1));
            if (spare) spare->push_back(p);
            return pair_name(p);
        }
        if (taken) *taken = found;
        return name;
    }

//------------------------------------------------------------------------------
//  Operands of a node.
//------------------------------------------------------------------------------
    struct operands {
        std::vector<std::string> text;
        std::vector<int> spare;         ///< register pairs that hold a second constant
    };
//  The texts of the operands `nodes` of one instruction (values brought into registers and waited for, constants
//  resolved); `negate[k]` flips the sign of operand k (a modifier for registers, the negated constant otherwise).
    operands resolve(const std::vector<uint32_t> &nodes, const std::vector<bool> &negate = std::vector<bool> ()) {
        operands out;
        for (const uint32_t o : nodes) {
            if (it.code[o].op != GFIR_CONST) pinned.insert(alias[o]);
        }
        for (const uint32_t o : nodes) {
            if (it.code[o].op != GFIR_CONST) fetch(alias[o]);
        }
        for (const uint32_t o : nodes) {
            if (it.code[o].op != GFIR_CONST) wait_for(values[static_cast<size_t> (alias[o])]);
        }
        int taken = -1;
        for (size_t k = 0; k < nodes.size(); k++) {
            const uint32_t o = nodes[k];
            const bool minus = k < negate.size() && negate[k];
            if (it.code[o].op == GFIR_CONST) {
                out.text.push_back(constant_operand(minus ? -it.code[o].imm[0] : it.code[o].imm[0], &taken, &out.spare));
            } else {
                out.text.push_back((minus ? "-" : "") + name_of(alias[o]));
            }
        }
        return out;
    }
    void done_with(operands &ops) {
        for (const int p : ops.spare) release_pair(p);
        ops.spare.clear();
    }

//  The value of node i now lives in pair p.
    void define(const size_t i, const int p) {
        value &v = values[i];
        v.defined = true;
        if (v.uses.empty()) {           // a node nobody reads (random items have a few): computed as the DAG says, kept nowhere
            release_pair(p);
            return;
        }
        v.reg = p;
        owner[static_cast<size_t> (p)] = static_cast<int64_t> (i);
    }

//  Values whose last reader was this position give their register (and LDS slot) back.
    void retire(const std::vector<int64_t> &read) {
        for (const int64_t id : read) {
            if (next_use(id, position + 1) != never) continue;
            value &v = values[static_cast<size_t> (id)];
            if (v.reg >= 0) {
//  (a parent's table value fetched ahead for a child that turned out to be in a register still: its load must have
//  landed before the pair takes another value)
                wait_for(v);
                release_pair(v.reg);
                v.reg = -1;
            }
            if (v.slot >= 0) {
                free_slots.push_back(v.slot);
                v.slot = -1;
            }
        }
    }

//------------------------------------------------------------------------------
//  Sequences (prelude.hpp: gf_rcp, gf_div, gf_sqrt_window, gf_pow_three_halves_window).
//------------------------------------------------------------------------------
    void reciprocal(const uint32_t d) {
        const int64_t id = static_cast<int64_t> (n + d);
        const std::string den = name_of(alias[d]);
        const int r = need_pair(), e = need_pair();
        const std::string R = pair_name(r), E = pair_name(e);
        vector_op("v_rcp_f64_e32 " + R + ", " + den);
        line("s_nop 0");
        vector_op("v_fma_f64 " + E + ", -" + den + ", " + R + ", 1.0");
        vector_op("v_fma_f64 " + R + ", " + R + ", " + E + ", " + R);
        vector_op("v_fma_f64 " + E + ", -" + den + ", " + R + ", 1.0");
        annotate("def " + value_name(id));
        vector_op("v_fma_f64 " + R + ", " + R + ", " + E + ", " + R);
        release_pair(e);
        values[static_cast<size_t> (id)].reg = r;
        values[static_cast<size_t> (id)].defined = true;
        owner[static_cast<size_t> (r)] = id;
    }

//  x -> sqrt(x) inside the window, into a new pair (returned); x stays where it is.
    int square_root(const std::string &x, const std::string &defines = std::string()) {
        const int g = need_pair(), h = need_pair(), y = need_pair();
        const std::string G = pair_name(g), H = pair_name(h), Y = pair_name(y);
        vector_op("v_rsq_f64_e32 " + Y + ", " + x);
        line("s_nop 0");
        vector_op("v_mul_f64 " + G + ", " + x + ", " + Y);
        vector_op("v_mul_f64 " + H + ", " + Y + ", 0.5");
        vector_op("v_fma_f64 " + Y + ", -" + H + ", " + G + ", 0.5");
        vector_op("v_fma_f64 " + G + ", " + G + ", " + Y + ", " + G);
        vector_op("v_fma_f64 " + H + ", " + H + ", " + Y + ", " + H);
        vector_op("v_fma_f64 " + Y + ", -" + G + ", " + G + ", " + x);
        vector_op("v_fma_f64 " + G + ", " + Y + ", " + H + ", " + G);
        vector_op("v_fma_f64 " + Y + ", -" + G + ", " + G + ", " + x);
        if (!defines.empty()) annotate(defines);
        vector_op("v_fma_f64 " + G + ", " + Y + ", " + H + ", " + G);
        release_pair(h);
        release_pair(y);
        return g;
    }

//  The argument of a sequence that tracks its high dword must sit in the pool: an input of the statement is copied.
    int in_pool(const uint32_t node, bool &copied) {
        value &v = values[static_cast<size_t> (alias[node])];
        copied = false;
        if (v.operand.empty()) return v.reg;
        const int p = need_pair();
        vector_op("v_mov_b64 " + pair_name(p) + ", " + v.operand);
        copied = true;
        return p;
    }

//  The cell of a gather group: clamp((x - offset)/scale) per dimension, through the shared sequence with the
//  literal reciprocal (codegen.hpp, index_expression), the quotients handed to the finite check (vmax).
    void find_cell(const size_t i) {
        const gfir_instruction &c = it.code[i];
        group &g = groups[static_cast<size_t> (node_group[i])];
        const table &t = it.tables[c.aux];
        const bool two = c.op == GFIR_GATHER2;
        const pack &pk = packs[g.pack];
        const int dimensions = two ? 2 : 1;
        int quotient[2] = {-1, -1}, index[2] = {-1, -1};
        for (int d = 0; d < dimensions; d++) {
            const uint32_t arg = d == 0 ? c.a : c.b;
            const double scale = c.imm[2*d], offset = c.imm[2*d + 1];
            operands x = resolve({arg});
            const int q = need_pair(), e = need_pair();
            const std::string Q = pair_name(q), E = pair_name(e);
            int taken = -1;
            if (it.code[arg].op == GFIR_CONST) {
//  (constant - offset) is not folded by the compiled body either: the subtraction is done here.
                vector_op("v_mov_b32_e32 " + low_name(e) + ", " + hex32(static_cast<uint32_t> (bits_of(it.code[arg].imm[0]))));
                vector_op("v_mov_b32_e32 " + high_name(e) + ", " + hex32(static_cast<uint32_t> (bits_of(it.code[arg].imm[0]) >> 32)));
                vector_op("v_add_f64 " + E + ", " + E + ", " + constant_operand(-offset, &taken));
            } else {
                vector_op("v_add_f64 " + E + ", " + x.text[0] + ", " + constant_operand(-offset, &taken));
            }
            done_with(x);
//  gf_div(n, scale, 1/scale):  q = n*r;  e' = fma(-scale, q, n);  q' = fma(e', r, q)   (n stays in E until e' needs its place)
            const int m = need_pair();
            const std::string M = pair_name(m);
            taken = -1;
            vector_op("v_mul_f64 " + Q + ", " + E + ", " + constant_operand(1.0/scale, &taken));
            taken = -1;
            vector_op("v_fma_f64 " + M + ", " + constant_operand(-scale, &taken) + ", " + Q + ", " + E);
            taken = -1;
            vector_op("v_fma_f64 " + Q + ", " + M + ", " + constant_operand(1.0/scale, &taken) + ", " + Q);
            release_pair(m);
            release_pair(e);
            quotient[d] = q;
        }
        {
            const std::string first = "abs(" + high_name(quotient[0]) + ")", second = "abs(" + high_name(quotient[dimensions - 1]) + ")";
            vector_op("v_maximum3_f32 %[vmax], %[vmax], " + first + ", " + second);
        }
        for (int d = 0; d < dimensions; d++) {
            const uint32_t length = d == 0 ? (two ? t.rows : t.cols) : t.cols;
            const std::string Q = pair_name(quotient[d]);
            int taken = -1;
            vector_op("v_max_f64 " + Q + ", " + Q + ", 0");
            vector_op("v_min_f64 " + Q + ", " + Q + ", " + constant_operand(static_cast<double> (length - 1), &taken));
            vector_op("v_cvt_u32_f64_e32 " + low_name(quotient[d]) + ", " + Q);
            index[d] = quotient[d];
        }
        const int cell = index[0];
        if (two) {
            if (t.cols <= 64) {
                vector_op("v_mad_u32_u24 " + low_name(cell) + ", " + low_name(index[0]) + ", " + std::to_string(t.cols) + ", " + low_name(index[1]));
            } else {
                vector_op("v_mul_u32_u24_e32 " + low_name(cell) + ", " + hex32(t.cols) + ", " + low_name(index[0]));
                vector_op("v_add_u32_e32 " + low_name(cell) + ", " + low_name(cell) + ", " + low_name(index[1]));
            }
            release_pair(index[1]);
        }
        const std::string defines = "def g" + std::to_string(node_group[i]) + " node " + std::to_string(i) + " stride " + std::to_string(pk.stride*8u) +
                                    " pack " + std::to_string(g.pack) + " lds " + std::to_string(pk.in_lds ? 1 : 0);
        if (!pk.in_lds) annotate(defines);
        vector_op("v_mul_u32_u24_e32 " + low_name(cell) + ", " + hex32(pk.stride*8u) + ", " + low_name(cell));
        if (pk.in_lds) {
            annotate(defines);
            vector_op("v_add_u32_e32 " + low_name(cell) + ", %[lds" + std::to_string(g.pack) + "], " + low_name(cell));
        }
        g.offset_pair = cell;
        owner[static_cast<size_t> (cell)] = -2;
    }

    void node(const size_t i) {
        const gfir_instruction &c = it.code[i];
        position = i;
        pinned.clear();
        look_ahead();
        std::vector<int64_t> read = used_at[i];
        switch (c.op) {
            case GFIR_CONST:
                return;
            case GFIR_INPUT:
                values[i].operand = "%[v" + std::to_string(c.a) + "]";
                values[i].defined = true;
                return;
            case GFIR_GATHER1:
            case GFIR_GATHER2:
                if (groups[static_cast<size_t> (node_group[i])].offset_pair == -1) find_cell(i);
                line("; alias r" + std::to_string(i) + " = " + value_name(alias[i]));
                break;
            case GFIR_ADD:
            case GFIR_SUB:
            case GFIR_MUL:
            case GFIR_FMA: {
                operands x = c.op == GFIR_FMA ? resolve({c.a, c.b, c.c}) : resolve({c.a, c.b}, {false, c.op == GFIR_SUB});
                retire(read);
                read.clear();
                const int p = need_pair();
                const char *name = c.op == GFIR_MUL ? "v_mul_f64 " : c.op == GFIR_FMA ? "v_fma_f64 " : "v_add_f64 ";
                std::string text = name + pair_name(p);
                for (auto &operand : x.text) text += ", " + operand;
                annotate("def r" + std::to_string(i));
                vector_op(text);
                done_with(x);
                define(i, p);
                break;
            }
            case GFIR_POWI: {
                operands x = resolve({c.a});
                const int p = need_pair();
                if (c.aux == 2) annotate("def r" + std::to_string(i));
                vector_op("v_mul_f64 " + pair_name(p) + ", " + x.text[0] + ", " + x.text[0]);
                for (uint32_t k = 2; k < c.aux; k++) {
                    if (k + 1 == c.aux) annotate("def r" + std::to_string(i));
                    vector_op("v_mul_f64 " + pair_name(p) + ", " + pair_name(p) + ", " + x.text[0]);
                }
                done_with(x);
                define(i, p);
                break;
            }
            case GFIR_DIV: {
                const int64_t r = static_cast<int64_t> (n + c.b);
                pinned.insert(r);
                operands x = resolve({c.a, c.b});
                if (!values[static_cast<size_t> (r)].defined) {
                    bool copied = false;
                    const int d = in_pool(c.b, copied);
                    track(d);
                    if (copied) {
                        flush_track();
                        release_pair(d);
                    }
                    reciprocal(c.b);
                } else {
                    fetch(r);
                    wait_for(values[static_cast<size_t> (r)]);
                }
                const std::string R = name_of(r);
                const int q = need_pair(), e = need_pair();
                const std::string Q = pair_name(q), E = pair_name(e);
                std::string negative_denominator = x.text[1][0] == '-' ? x.text[1].substr(1) : "-" + x.text[1];
                vector_op("v_mul_f64 " + Q + ", " + x.text[0] + ", " + R);
                vector_op("v_fma_f64 " + E + ", " + negative_denominator + ", " + Q + ", " + x.text[0]);
                annotate("def r" + std::to_string(i));
                vector_op("v_fma_f64 " + Q + ", " + E + ", " + R + ", " + Q);
                release_pair(e);
                done_with(x);
                define(i, q);
                break;
            }
            case GFIR_SQRT: {
                operands x = resolve({c.a});
                bool copied = false;
                const int argument = in_pool(c.a, copied);
                track(argument);
                const int p = square_root(pair_name(argument), "def r" + std::to_string(i));
                if (copied) {
                    flush_track();
                    release_pair(argument);
                }
                done_with(x);
                define(i, p);
                break;
            }
            case GFIR_POW: {
//  gf_pow_three_halves_window:  s = sqrt(x);  t = fma(-s, s, x)*(0.5*rcp(s));  p = x*s;  c = fma(x, s, -p) + x*t;  p + c
                operands x = resolve({c.a});
                bool copied = false;
                const int argument = in_pool(c.a, copied);
                track(argument);
                const std::string X = pair_name(argument);
                const int s = square_root(X);
                const int t = need_pair(), u = need_pair(), p = need_pair();
                const std::string S = pair_name(s), T = pair_name(t), U = pair_name(u), P = pair_name(p);
                vector_op("v_rcp_f64_e32 " + T + ", " + S);
                line("s_nop 0");
                vector_op("v_mul_f64 " + T + ", " + T + ", 0.5");
                vector_op("v_fma_f64 " + U + ", -" + S + ", " + S + ", " + X);
                vector_op("v_mul_f64 " + T + ", " + U + ", " + T);
                vector_op("v_mul_f64 " + P + ", " + X + ", " + S);
                vector_op("v_fma_f64 " + U + ", " + X + ", " + S + ", -" + P);
                vector_op("v_mul_f64 " + T + ", " + X + ", " + T);
                vector_op("v_add_f64 " + U + ", " + U + ", " + T);
                annotate("def r" + std::to_string(i));
                vector_op("v_add_f64 " + P + ", " + P + ", " + U);
                release_pair(s);
                release_pair(t);
                release_pair(u);
                if (copied) {
                    flush_track();
                    release_pair(argument);
                }
                done_with(x);
                define(i, p);
                break;
            }
            default:
                result.why = "an operation without a sequence in asm_body.hpp";
                return;
        }
        retire(read);
//  A group's cell offset is given back after the last read of any of its table values.
        for (auto &g : groups) {
            if (g.offset_pair >= 0 && g.last_use <= i) {
                release_pair(g.offset_pair);
                g.offset_pair = -2;
            }
        }
    }

  public:
    asm_body_text write() {
        result.why = why_not(it, opt);
        if (!result.why.empty()) return result;
        analyse();
        owner.assign(pairs, -1);
        line("s_waitcnt lgkmcnt(0)");
        for (size_t i = 0; i < n && result.why.empty(); i++) {
            node(i);
            if (std::getenv("GFHIP_ASM_TRACE") && (i%100 == 0 || !result.why.empty())) {
                size_t in_registers = 0, tables = 0, reciprocals = 0, parked = 0;
                for (size_t p = 0; p < owner.size(); p++) {
                    if (owner[p] == -1) continue;
                    in_registers++;
                    if (owner[p] >= static_cast<int64_t> (2*n)) tables++;
                    else if (owner[p] >= static_cast<int64_t> (n)) reciprocals++;
                }
                for (auto &v : values) parked += v.slot >= 0 && v.reg < 0;
                std::fprintf(stderr, "  node %zu: %zu pairs in use (%zu table values, %zu reciprocals), %zu values only in LDS, %zu slots taken\n",
                             i, in_registers, tables, reciprocals, parked, static_cast<size_t> (result.slots) - free_slots.size());
            }
        }
        if (!result.why.empty()) return result;
//  The stored values: from the pool (or from an input) into the statement's outputs, inputs first — an output may
//  share its registers with an input the compiler knows to be dead by then.
        position = n;
        pinned.clear();
        flush_track();
        std::vector<std::pair<std::string, uint32_t>> targets;
        for (size_t k = 0; k < it.setters.size(); k++) targets.push_back({"%[sv" + std::to_string(k) + "]", it.setters[k].value});
        for (size_t o = 0; o < it.outputs.size(); o++) targets.push_back({"%[so" + std::to_string(o) + "]", it.outputs[o]});
        std::map<uint32_t, int> staged;
        for (auto &target : targets) {
            const gfir_instruction &c = it.code[target.second];
            if (c.op == GFIR_CONST) continue;
            const int64_t id = alias[target.second];
            pinned.insert(id);
            if (!values[static_cast<size_t> (id)].operand.empty() && !staged.count(target.second)) {
                const int p = need_pair();
                vector_op("v_mov_b64 " + pair_name(p) + ", " + values[static_cast<size_t> (id)].operand);
                staged[target.second] = p;
            }
        }
        for (auto &target : targets) {
            const gfir_instruction &c = it.code[target.second];
            if (c.op == GFIR_CONST) {
                const uint64_t bits = bits_of(c.imm[0]);
                const int p = need_pair();
                vector_op("v_mov_b32_e32 " + low_name(p) + ", " + hex32(static_cast<uint32_t> (bits)));
                vector_op("v_mov_b32_e32 " + high_name(p) + ", " + hex32(static_cast<uint32_t> (bits >> 32)));
                annotate("out constant " + std::to_string(bits));
                vector_op("v_mov_b64 " + target.first + ", " + pair_name(p));
                release_pair(p);
                continue;
            }
            const int64_t id = alias[target.second];
            if (staged.count(target.second)) {
                annotate("out r" + std::to_string(target.second));
                vector_op("v_mov_b64 " + target.first + ", " + pair_name(staged[target.second]));
                continue;
            }
            fetch(id);
            wait_for(values[static_cast<size_t> (id)]);
            annotate("out r" + std::to_string(target.second));
            vector_op("v_mov_b64 " + target.first + ", " + name_of(id));
        }
        line("s_waitcnt vmcnt(0) lgkmcnt(0)");
        if (!result.why.empty()) return result;

//  Operands and clobbers.
        std::ostringstream s;
        s << "                asm volatile(\n" << a.str() << "                : ";
        bool first = true;
        auto separator = [&] () { if (!first) s << ", "; first = false; };
        for (size_t k = 0; k < it.setters.size(); k++) { separator(); s << "[sv" << k << "] \"=v\"(sv" << k << ")"; }
        for (size_t o = 0; o < it.outputs.size(); o++) { separator(); s << "[so" << o << "] \"=v\"(so" << o << ")"; }
        separator();
        s << "[dmax] \"+v\"(dmax), [dmin] \"+v\"(dmin), [vmax] \"+v\"(vmax)\n                : ";
        first = true;
        std::vector<bool> input_read(it.symbols.size(), false);
        for (size_t i = 0; i < n; i++) {
            if (it.code[i].op == GFIR_INPUT && !values[i].uses.empty()) input_read[it.code[i].a] = true;
        }
        for (size_t i = 0; i < it.symbols.size(); i++) {
            if (input_read[i]) { separator(); s << "[v" << i << "] \"v\"(v" << i << ")"; }
        }
        for (size_t p = 0; p < packs.size(); p++) {
            separator();
            if (packs[p].in_lds) s << "[lds" << p << "] \"v\"(lds_address" << p << ")";
            else s << "[pack" << p << "] \"s\"(pack" << p << ")";
        }
        if (result.slots) {
            separator();
            const uint32_t per_base = 65536u/(block_size*8u);
            for (uint32_t k = 0; k*per_base < result.slots; k++) {
                if (k) s << ", ";
                s << "[park" << k << "] \"v\"(park" << k << ")";
            }
        }
        s << "\n                : ";
        for (uint32_t r = pool_lo; r < 256; r++) s << "\"v" << r << "\", ";
        for (int r = sgpr_lo; r < sgpr_lo + 2*sgpr_pairs; r++) s << "\"s" << r << "\", ";
        s << "\"memory\");\n";
        result.statement = s.str();
        result.ok = true;
        return result;
    }
};

//------------------------------------------------------------------------------
///  @brief The emission order for the assembly body: of `tries` tie-breaks of the list schedule, the one whose
///  body needs the fewest LDS slots (then the fewest LDS round trips).
///
///  The statement is written for every candidate (a millisecond or two each for the RK4 item) with the slot limit
///  lifted; with two waves per SIMD a lane has 40 slots at most, the RK4 item needs 19 with its best order and 65
///  with the order schedule_for_pressure picks for the compiler.
//------------------------------------------------------------------------------
///  `directories`: where the seed of the chosen order is remembered (`<hash of item and knobs>.order`, next to the code
///  objects of the kernel cache): the search then runs once per item and machine, not once per process.
inline item schedule_for_assembly(const item &in, const codegen_options &opt, const std::vector<std::string> &directories = {}) {
    if (in.code.size() > 20000 || !asm_body_writer::why_not(in, opt).empty()) return schedule_for_pressure(in);
    if (const char *forced = std::getenv("GFHIP_ASM_SEED")) {         // experiment (profiles/diag/asm/seeds.sh): this tie-break, no search
        return reorder(in, list_schedule(in, static_cast<uint32_t> (std::atoi(forced))));
    }
    const std::vector<uint8_t> bytes = in.serialize();
    const uint64_t key = fnv1a(std::string(bytes.begin(), bytes.end()) + "|order|" + std::to_string(opt.asm_pool_lo) + "|" +
                               std::to_string(opt.asm_load_ahead) + "|" + std::to_string(opt.asm_reload_ahead) + "|" +
                               std::to_string(opt.asm_schedule_tries) + "|" + std::to_string(opt.asm_wide_loads) + "|" +
                               std::to_string(opt.lds_budget) + "|" + std::to_string(opt.compact_tables) + "|" + std::to_string(opt.block_size));
    char name[40];
    std::snprintf(name, sizeof(name), "/%016llx.order", static_cast<unsigned long long> (key));
    for (auto &directory : directories) {
        if (FILE *f = std::fopen((directory + name).c_str(), "r")) {
            unsigned int seed = 0;
            const bool read = std::fscanf(f, "%u", &seed) == 1;
            std::fclose(f);
            if (read) return reorder(in, list_schedule(in, seed));
        }
    }
    const table_layout layout = layout_tables(in, opt);
    item best;
    uint32_t best_seed = 0;
    size_t best_slots = ~static_cast<size_t> (0), best_traffic = 0;
//  (the search costs tries x one writing of the statement: fewer tries for items far larger than the RK4 step)
    const uint32_t tries = in.code.size() <= 5000 ? opt.asm_schedule_tries
                         : std::max<uint32_t> (4u, static_cast<uint32_t> (opt.asm_schedule_tries*5000ull/in.code.size()));
    for (uint32_t seed = 0; seed < std::max(1u, tries); seed++) {
        item candidate = reorder(in, list_schedule(in, seed));
        asm_body_writer writer(candidate, opt, layout.packs, layout.parent, layout.factor, layout.table_pack, layout.table_column,
                               opt.block_size, 1u << 20);
        const asm_body_text text = writer.write();
        if (!text.ok) continue;
        const size_t traffic = text.lds_reads + text.lds_writes;
        if (std::getenv("GFHIP_ASM_REPORT")) {
            std::fprintf(stderr, "  order %u of %s: %u slots, %zu LDS reads, %zu LDS writes, %zu loads, %zu waits\n", seed, in.name.c_str(), text.slots,
                         text.lds_reads, text.lds_writes, text.loads, text.waits);
        }
        if (text.slots < best_slots || (text.slots == best_slots && traffic < best_traffic)) {
            best_slots = text.slots;
            best_traffic = traffic;
            best = std::move(candidate);
            best_seed = seed;
        }
    }
    if (best.code.empty()) return schedule_for_pressure(in);
    for (auto &directory : directories) {
        if (FILE *f = std::fopen((directory + name).c_str(), "w")) {
            std::fprintf(f, "%u\n", best_seed);
            std::fclose(f);
            break;
        }
    }
    return best;
}

///  Whether the statement of `ordered` can be written within the LDS a workgroup has (lower() decides the same way).
inline bool assembly_fits(const item &ordered, const codegen_options &opt) {
    const table_layout layout = layout_tables(ordered, opt);
    const size_t per_block = 160u*1024u/opt.asm_waves;
    const uint32_t slot_limit = layout.lds_used < per_block
                              ? static_cast<uint32_t> ((per_block - layout.lds_used)/(static_cast<size_t> (opt.block_size)*ordered.element_size())) : 0;
    asm_body_writer writer(ordered, opt, layout.packs, layout.parent, layout.factor, layout.table_pack, layout.table_column, opt.block_size, slot_limit);
    return writer.write().ok;
}

}  // namespace gfhip

#endif /* gfhip_asm_body_hpp */

//------------------------------------------------------------------------------
///  @file options.hpp
///  @brief Knobs of the lowering (defaults are the measured best; every override changes the
///  generated text and therefore the kernel-cache key), the cache hash and the compile flags.
///
///  Alternatives that were measured neutral or slower on MI355X (packed fp32 pairs, 2/4 rays per
///  lane, next-tile prefetch, pipelined tiles, scheduling fences, explicit order files) are
///  recorded in DESIGN.md and are no longer part of the lowering.
//------------------------------------------------------------------------------
#ifndef gfhip_options_hpp
#define gfhip_options_hpp

#include <cstdint>
#include <cstdlib>
#include <string>

namespace gfhip {

///  How a division node is computed (all three give the IEEE quotient; see prelude.hpp).
enum class division_mode {
    shared,     ///< reciprocal shared per denominator; lanes that leave the checked window redo the pass with `ieee`
    checked,    ///< `shared` + every numerator's magnitude is tracked as well (complete for tiny numerators too)
    ieee,       ///< the compiler's own division sequence for every quotient
    fast        ///< NOT bit-exact, opt-in: q = n*r with the refined shared reciprocal and no residual step (<= ~1.5 ulp per quotient),
                ///< no checks, no second body.  north_star's bound is 1e-6 relative on the trajectories; the tests hold this mode to it.
};

struct codegen_options {
    size_t lds_budget = 64*1024;        ///< bytes of LDS the staged packs may use per workgroup (GFHIP_LDS_BUDGET)
    uint32_t block_size = 256;
    uint32_t waves_per_simd = 0;        ///< second __launch_bounds__ argument, 0 = the compiler decides (GFHIP_WAVES_PER_SIMD)
    division_mode division = division_mode::shared;     ///< GFHIP_DIVISION=shared|checked|ieee
    bool pow_three_halves = true;       ///< fp64 pow(x, 1.5) as a compensated x*sqrt(x) (GFHIP_POW=libm: ocml's pow)
    bool window_sqrt = false;           ///< fp64 sqrt and pow(x, 1.5) inside the checked window as the 9-instruction core of the compiler's
                                        ///< 18-instruction sqrt (same bits, GFHIP_WINDOW_SQRT=1).  Measured on the RK4 kernel: 809 fewer static
                                        ///< but 2 % MORE executed vector instructions per pass (SQ_INSTS_VALU), 84 more AGPR copies, 1.5 %
                                        ///< SLOWER (2.077 vs 2.045 ms at 1e7 rays): off by default
    bool window_sqrt_f32 = true;        ///< the same for sqrtf in fp32 items (16 -> 9 instructions, GFHIP_WINDOW_SQRT_F32=0 turns it off)
    int nontemporal = -1;               ///< nt hint on the state loads and stores: -1 = items of 100 nodes and more (measured: xkorc push
                                        ///< -5.5 %, Newton item -2.7 % at 1e7 elements, neutral at 1e6; items that only stream lose),
                                        ///< 0 = never, 1 = always, 2 = stores only, 3 = loads only (GFHIP_NONTEMPORAL)
    bool compact_tables = true;         ///< store only tables that are not an exact multiple of another (GFHIP_COMPACT_TABLES=0)
    bool park_in_lds = true;            ///< very long-lived values wait in LDS instead of AGPRs/scratch (GFHIP_PARK=0|1|heavy)
    uint32_t park_min_range = 1500;     ///< park values whose live range exceeds this many nodes (heavy: 300) ...
    uint32_t park_window = 100;         ///< ... uses closer than this share one reload
    uint32_t park_max_slots = 32;       ///< LDS slots of block_size elements each
    uint32_t park_prefetch = 50;        ///< issue a reload this many nodes before its first use (< window)
    uint32_t park_capacity = 0;         ///< > 0: parking planned by a Belady replay with this many fp64 values in registers (GFHIP_PARK_CAPACITY)
    bool schedule_for_pressure = true;  ///< emit in the pressure-aware order of schedule.hpp (GFHIP_SCHEDULE=source: item order)
    int division_fixup = -1;            ///< v_div_fixup after each shared-reciprocal quotient: 1 yes, 0 no, -1 auto (GFHIP_DIV_FIXUP)
    uint32_t converge_batch = 3;        ///< converge items: passes per launch of `<name>_batch` (state in registers between them, one max per
                                        ///< pass; GFHIP_CONVERGE_BATCH, 1 = one launch per pass)
    uint32_t segment_nodes = 6000;      ///< items of more records are cut into segments of about this many, each a kernel of its own
                                        ///< (segments.hpp; GFHIP_SEGMENT_NODES, 0 = never)
    uint32_t segments = 0;              ///< experiment: cut every item of `segments_min_nodes` records and more into this many segments (GFHIP_SEGMENTS)
    uint32_t segments_min_nodes = 2000; ///< ... (GFHIP_SEGMENTS_MIN_NODES; the tests split small items to reach the redo launch)
    bool asm_body = true;               ///< fp64 items of `asm_min_nodes` records and more: the body of a pass as gfx950 assembly with a register
                                        ///< assignment of its own (asm_body.hpp; two waves per SIMD, no AGPR copies; GFHIP_ASM=0: the compiled
                                        ///< body).  RK4 item, 1e7 rays: 1.814 ms per step against 2.043 ms
    uint32_t asm_waves = 2;             ///< ... waves per SIMD the kernel is built for: 2 leaves each workgroup 80 KB of LDS (40 slots) (GFHIP_ASM_WAVES)
    bool asm_wide_loads = true;         ///< ... two neighbouring table columns wanted soon come with one 16-byte load (GFHIP_ASM_WIDE_LOADS=0)
    uint32_t asm_schedule_tries = 64;   ///< ... tie-breaks of the list schedule tried for the order that needs the fewest LDS slots (GFHIP_ASM_TRIES)
    uint32_t asm_min_nodes = 1000;      ///< ... (GFHIP_ASM_MIN_NODES)
    uint32_t asm_pool_lo = 40;          ///< first VGPR of the assembly body's pool; the compiler keeps v0..v(lo-1) (GFHIP_ASM_POOL_LO)
    uint32_t asm_load_ahead = 96;       ///< table loads are issued this many nodes ahead of their first use (GFHIP_ASM_LOAD_AHEAD) ...
    uint32_t asm_reload_ahead = 24;     ///< ... LDS reads (values sent out of the registers, LDS-staged tables) this many (GFHIP_ASM_RELOAD_AHEAD).
                                        ///< Measured (profiles/r03_asm_sweep.jsonl): 0/0 2.19 ms, 24/6 1.88, 48/12 1.85, 96/24 1.82; pool from v48: 1.83
    size_t handover_bytes = 128u << 20; ///< the hand-over buffers of a segmented item hold one chunk of rays and at most this many
                                        ///< bytes, so that they stay in the 256 MB Infinity Cache (GFHIP_HANDOVER_BYTES)

//  Environment overrides (they change the generated text, hence the cache key).
    static codegen_options from_environment() {
        codegen_options o;
        if (const char *e = std::getenv("GFHIP_DIVISION")) {
            const std::string mode(e);
            o.division = mode == "ieee" ? division_mode::ieee : mode == "checked" ? division_mode::checked
                       : mode == "fast" ? division_mode::fast : division_mode::shared;
        }
        if (const char *e = std::getenv("GFHIP_PARK")) {
            const std::string mode(e);
            o.park_in_lds = mode != "0";
            if (mode == "heavy") o.park_min_range = 300;
        }
        if (const char *e = std::getenv("GFHIP_SCHEDULE")) o.schedule_for_pressure = std::string(e) != "source";
        if (const char *e = std::getenv("GFHIP_DIV_FIXUP")) o.division_fixup = std::string(e) != "0" ? 1 : 0;
        if (const char *e = std::getenv("GFHIP_NONTEMPORAL")) o.nontemporal = std::atoi(e);
        if (const char *e = std::getenv("GFHIP_COMPACT_TABLES")) o.compact_tables = std::string(e) != "0";
        if (const char *e = std::getenv("GFHIP_POW")) o.pow_three_halves = std::string(e) != "libm";
        if (const char *e = std::getenv("GFHIP_PARK_CAPACITY")) o.park_capacity = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_PARK_PREFETCH")) o.park_prefetch = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_ASM")) o.asm_body = std::string(e) != "0";
        if (const char *e = std::getenv("GFHIP_ASM_WAVES")) o.asm_waves = static_cast<uint32_t> (std::atoi(e)) == 2 ? 2 : 1;
        if (const char *e = std::getenv("GFHIP_ASM_WIDE_LOADS")) o.asm_wide_loads = std::string(e) != "0";
        if (const char *e = std::getenv("GFHIP_ASM_TRIES")) o.asm_schedule_tries = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_ASM_MIN_NODES")) o.asm_min_nodes = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_ASM_POOL_LO")) o.asm_pool_lo = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_ASM_LOAD_AHEAD")) o.asm_load_ahead = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_ASM_RELOAD_AHEAD")) o.asm_reload_ahead = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_WINDOW_SQRT")) o.window_sqrt = std::string(e) == "1";
        if (const char *e = std::getenv("GFHIP_WINDOW_SQRT_F32")) o.window_sqrt_f32 = std::string(e) == "1";
        if (const char *e = std::getenv("GFHIP_WAVES_PER_SIMD")) o.waves_per_simd = static_cast<uint32_t> (std::atoi(e));
        if (const char *e = std::getenv("GFHIP_LDS_BUDGET")) o.lds_budget = static_cast<size_t> (std::atol(e));
        if (const char *e = std::getenv("GFHIP_CONVERGE_BATCH")) o.converge_batch = static_cast<uint32_t> (std::atol(e) > 0 ? std::atol(e) : 1);
        if (o.converge_batch > 8) o.converge_batch = 8;
        if (const char *e = std::getenv("GFHIP_SEGMENT_NODES")) o.segment_nodes = static_cast<uint32_t> (std::atol(e));
        if (const char *e = std::getenv("GFHIP_SEGMENTS")) o.segments = static_cast<uint32_t> (std::atol(e));
        if (const char *e = std::getenv("GFHIP_SEGMENTS_MIN_NODES")) o.segments_min_nodes = static_cast<uint32_t> (std::atol(e));
        if (const char *e = std::getenv("GFHIP_HANDOVER_BYTES")) o.handover_bytes = static_cast<size_t> (std::atoll(e));
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
    return "-O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950";
}

}  // namespace gfhip

#endif /* gfhip_options_hpp */

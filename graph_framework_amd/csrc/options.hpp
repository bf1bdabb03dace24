//------------------------------------------------------------------------------
///  @file options.hpp
///  @brief Knobs of the lowering (defaults are the measured best; every override changes the
///  generated text and therefore the kernel-cache key), the cache hash and the compile flags.
//------------------------------------------------------------------------------
#ifndef gfhip_options_hpp
#define gfhip_options_hpp

#include <cstdint>
#include <cstdlib>
#include <string>

namespace gfhip {

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

}  // namespace gfhip

#endif /* gfhip_options_hpp */

// Host-only driver for the GFIR parser, scheduler and lowering (no HIP runtime): built with
// -fsanitize=address,undefined by tests/test_cabi.py and run over every exported workload and
// over mutated items; every item that can be split is also cut into 2..5 segments (csrc/segments.hpp), each segment
// serialized, parsed again and lowered in the roles of a split with a redo launch.
// Usage: lowering_sanitize <file.gfir>... [--mutate seed trials file.gfir]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>

#include "../include/gfir.h"
#include "../graph_framework_amd/csrc/codegen.hpp"
#include "../graph_framework_amd/csrc/segments.hpp"

static std::vector<char> read_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<char> ((std::istreambuf_iterator<char> (f)), std::istreambuf_iterator<char> ());
}

static bool lower_bytes(const std::vector<char> &bytes, uint64_t &hash) {
    gfhip::item it;
    std::string error;
    if (!it.parse(bytes.data(), bytes.size(), error)) return false;
    const gfhip::lowered low = gfhip::lower(it);
    hash = low.hash;
    if (gfhip::can_split(it) && it.code.size() >= 40 && it.code.size() < 20000) {
        const gfhip::item ordered = gfhip::schedule_for_pressure(it);
        for (size_t count = 2; count <= 5; count++) {
            gfhip::segmentation plan = gfhip::split_item(ordered, gfhip::choose_cuts(ordered, count));
            for (size_t p = 0; p < plan.segments.size(); p++) {
                const std::vector<uint8_t> blob = plan.segments[p].piece.serialize();
                gfhip::item again;
                if (!again.parse(blob.data(), blob.size(), error)) {
                    std::fprintf(stderr, "a segment does not parse: %s\n", error.c_str());
                    std::exit(1);
                }
                gfhip::piece_info role;
                role.role = p + 1 == plan.segments.size() ? gfhip::piece_role::last : gfhip::piece_role::middle;
                for (auto slot : plan.segments[p].output_slot) role.output_handed_over.push_back(slot >= 0);
                hash ^= gfhip::lower(again, gfhip::codegen_options(), role).hash;
            }
        }
//  The whole item as one piece whose pass is the assembly body (csrc/asm_body.hpp), whatever its size: the order search
//  (three tie-breaks here), the writer with its register pool, LDS slots and look-ahead, a small pool as well.
        for (const uint32_t pool : {40u, 200u}) {
            gfhip::codegen_options assembly;
            assembly.asm_min_nodes = 0;
            assembly.asm_schedule_tries = 3;
            assembly.asm_pool_lo = pool;
            assembly.asm_waves = 1;
            const gfhip::item chosen = gfhip::schedule_for_assembly(it, assembly);
            gfhip::piece_info whole;
            whole.role = gfhip::piece_role::last;
            whole.scheduled = true;
            hash ^= gfhip::lower(chosen, assembly, whole).hash;
        }
        gfhip::piece_info redo;
        redo.role = gfhip::piece_role::redo;
        gfhip::codegen_options plain;
        plain.division = gfhip::division_mode::ieee;
        hash ^= gfhip::lower(it, plain, redo).hash;
    }
    return true;
}

int main(int argc, char **argv) {
    size_t lowered = 0, rejected = 0;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--mutate") && i + 3 < argc) {
            std::mt19937_64 rng(std::strtoull(argv[i + 1], nullptr, 10));
            const size_t trials = std::strtoull(argv[i + 2], nullptr, 10);
            const std::vector<char> base = read_file(argv[i + 3]);
            for (size_t t = 0; t < trials; t++) {
                std::vector<char> b = base;
                const unsigned kind = rng()%10;
                if (kind < 3) {
                    b.resize(rng()%b.size());
                } else if (kind < 8) {
                    for (unsigned k = 0, n = 1 + rng()%5; k < n; k++) b[rng()%b.size()] = static_cast<char> (rng());
                } else {
                    const uint32_t values[5] = {0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 100000u, static_cast<uint32_t> (rng())};
                    const uint32_t v = values[rng()%5];
                    std::memcpy(b.data() + (rng()%(b.size()/4))*4, &v, 4);
                }
                uint64_t hash;
                (lower_bytes(b, hash) ? lowered : rejected)++;
            }
            i += 3;
            continue;
        }
        uint64_t hash = 0;
        if (!lower_bytes(read_file(argv[i]), hash)) {
            std::fprintf(stderr, "%s: rejected\n", argv[i]);
            return 1;
        }
        lowered++;
    }
    std::printf("lowered %zu rejected %zu\n", lowered, rejected);
    return 0;
}

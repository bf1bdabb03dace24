//------------------------------------------------------------------------------
///  @file gf_hip.cpp
///  @brief libgf_hip.so: the C ABI of include/gf_hip.h on the HIP runtime.
///
///  Host side of the MI355X backend: owns the device buffers (keyed like the
///  reference's kernel_arguments maps, cuda_context.hpp:78-80), lowers GFIR work
///  items with codegen.hpp, builds them (cached code objects or hipRTC), packs
///  and uploads the coefficient tables, launches on one HIP stream per context
///  and runs the converge loop of workflow.hpp:179-205 around the device max
///  reduction of reduce.hip.
//------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gf_hip.h"
#include "codegen.hpp"
#include "segments.hpp"

#include "converge_state.hpp"

namespace gfhip {
void launch_max_reduce(const void *in, const size_t n, const bool f64,
                       unsigned long long *result, const unsigned int num_cus, hipStream_t stream);
void launch_converge_decide(const bool f64, unsigned long long *reduced, void *state, hipStream_t stream);
void launch_converge_decide_batch(const bool f64, unsigned long long *reduced, void *state, const unsigned int count, hipStream_t stream);
void launch_max_modulus(const void *in, const size_t n, const bool f64, void *result, hipStream_t stream);
}

namespace {

thread_local std::string creation_error;

struct buffer {
    void *pointer = nullptr;
    size_t count = 0;
    uint32_t dtype = GFIR_F64;
    bool owned = true;
    void *mirror = nullptr;     // pinned host copy handed out by gfhip_get_host_buffer, refreshed by gfhip_wait
};

std::string library_directory() {
    Dl_info info;
    if (dladdr(reinterpret_cast<void *> (&gfhip_max_concurrency), &info) && info.dli_fname) {
        std::string path(info.dli_fname);
        const size_t slash = path.rfind('/');
        return slash == std::string::npos ? "." : path.substr(0, slash);
    }
    return ".";
}

bool read_file(const std::string &path, std::vector<char> &data) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    const std::streamoff size = f.tellg();
    f.seekg(0, std::ios::beg);
    data.resize(static_cast<size_t> (size));
    f.read(data.data(), size);
    return static_cast<bool> (f);
}

size_t element_bytes(const uint32_t dtype) {
    return gfhip::item::element_size(dtype);
}

std::string hash_name(const uint64_t hash) {
    char buf[32];
    std::snprintf(buf, sizeof(buf), "%016llx", static_cast<unsigned long long> (hash));
    return buf;
}

}  // namespace

struct gfhip_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    unsigned int num_cus = 256;
    std::map<uint64_t, buffer> buffers;
    std::map<uint64_t, std::pair<void *, size_t>> random_states;    // MT19937 states per random_state node (raw bytes)
    std::vector<std::unique_ptr<gfhip_kernel>> kernels;
    std::string error;
    unsigned long long *device_scalar = nullptr;
    unsigned long long *host_scalar = nullptr;     // pinned, 8 words
    gfhip_kernel *running_ahead = nullptr;         // the kernel whose last batch ran passes the caller has not asked for yet
    gfhip_kernel *max_streak = nullptr;            // the kernel the last entry point was gfhip_run_max of
    unsigned int *device_flags = nullptr;          // bit 0: a lane redid a pass with the compiler's division
    gfhip::converge_state *device_converge = nullptr;
    gfhip::converge_state *host_converge = nullptr;   // pinned
    unsigned int timing = 0;                       // 0 = off, N = events around every Nth launch of a kernel

    int fail(const std::string &message) {
        error = message;
        return 1;
    }
    int check(const hipError_t status, const char *what) {
        if (status != hipSuccess) {
            return fail(std::string(what) + ": " + hipGetErrorString(status));
        }
        return 0;
    }
};

//  One compiled kernel of a work item: the item itself, or one segment of it (segments.hpp).
struct built_piece {
    gfhip::segment plan;                           // the piece as an item, and what its symbols and outputs are
    gfhip::lowered low;
    hipModule_t module = nullptr;
    hipFunction_t function = nullptr;
    std::vector<void *> pack_device;
    unsigned int grid = 1;
    int vgprs = 0, lds_static = 0, scratch = 0;
    bool from_cache = false;
};

struct gfhip_kernel {
    gfhip_context *ctx = nullptr;
    gfhip::item item;
    gfhip::lowered low;
    std::vector<built_piece> pieces;               // non-empty: the item runs as this sequence of segment kernels
    std::vector<void *> handover;                  // one device array of `chunk` elements per hand-over slot
    size_t chunk = 0;                              // rays per walk of the segment sequence
    bool has_redo = false;                         // lanes outside the division window are redone by `redo`, a launch of its own
    built_piece redo;                              // the whole item with the compiler's division, over the redo list
    unsigned char *flagged = nullptr;              // per ray: a segment before the last found it outside the window
    unsigned int *redo_list = nullptr, *redo_count = nullptr;
    size_t num_rays = 0;
    hipModule_t module = nullptr;
    hipFunction_t function = nullptr;
    hipFunction_t converge_function = nullptr;
    hipFunction_t max_function = nullptr;
    hipFunction_t batch_function = nullptr;         // `<name>_batch`: several passes per launch, one max per pass
    std::vector<void *> undo;                      // per setter: the target's values at the beginning of the last batch
//  gfhip_run_max called in a row (the reference's converge_item::run, workflow.hpp:179-205, through hip_context's
//  create_max_call): passes of the last `<name>_batch` launch that ran ahead of the caller, their maxes waiting here.
    std::vector<double> ahead;
    unsigned int ahead_taken = 0;
    bool built = false;
    bool from_cache = false;
    std::vector<void *> pack_device;
    std::vector<uint64_t> input_keys, output_keys;
    void *random_states = nullptr;                 // device copy of the item's random_state node (items with draws)
    bool bound = false;
    unsigned int grid = 1;
    int vgprs = 0, sgprs = 0, lds_static = 0, scratch = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    uint64_t launch_count = 0;
    std::vector<double> samples;                   // durations drained by the last gfhip_kernel_timing
};

#define GFHIP_TRY(ctx, call, what) do { if ((ctx)->check((call), (what))) return 1; } while (0)

extern "C" int gfhip_max_concurrency(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        return 0;
    }
    return count;
}

extern "C" const char *gfhip_device_type(void) {
    return "HIP GPU";
}

extern "C" int gfhip_shard_bounds(size_t total, size_t shards, size_t index, size_t *begin, size_t *end) {
    if (shards == 0 || index >= shards) return 1;
    const size_t batch = total/shards;
    const size_t extra = total%shards;
    const size_t first = index*batch + (index < extra ? index : extra);
    if (begin) *begin = first;
    if (end) *end = first + batch + (extra > index ? 1 : 0);
    return 0;
}

extern "C" const char *gfhip_last_error(const gfhip_context *ctx) {
    return ctx ? ctx->error.c_str() : creation_error.c_str();
}

extern "C" gfhip_context *gfhip_create_context(int index, void *stream) {
    int count = 0;
    hipError_t status = hipGetDeviceCount(&count);
    if (status != hipSuccess || count == 0) {
        creation_error = "no HIP device available";
        return nullptr;
    }
    if (index < 0 || index >= count) {
        creation_error = "device index out of range";
        return nullptr;
    }
    std::unique_ptr<gfhip_context> ctx(new gfhip_context);
    ctx->device = index;
    if ((status = hipSetDevice(index)) != hipSuccess) {
        creation_error = std::string("hipSetDevice: ") + hipGetErrorString(status);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, index) == hipSuccess) {
        ctx->num_cus = static_cast<unsigned int> (prop.multiProcessorCount);
        const std::string arch(prop.gcnArchName);
        if (arch.rfind("gfx950", 0) != 0) {
            creation_error = "libgf_hip is built for gfx950 (MI355X); device is " + arch;
            return nullptr;
        }
    }
    if (stream) {
        ctx->stream = static_cast<hipStream_t> (stream);
    } else {
        if ((status = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
            creation_error = std::string("hipStreamCreate: ") + hipGetErrorString(status);
            return nullptr;
        }
        ctx->own_stream = true;
    }
    if (hipMalloc(reinterpret_cast<void **> (&ctx->device_flags), sizeof(unsigned int)) != hipSuccess ||
        hipMemset(ctx->device_flags, 0, sizeof(unsigned int)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **> (&ctx->device_scalar), 8*sizeof(unsigned long long)) != hipSuccess ||     // one per pass of a batch
        hipMemset(ctx->device_scalar, 0, 8*sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **> (&ctx->host_scalar), 8*sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **> (&ctx->device_converge), sizeof(gfhip::converge_state)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **> (&ctx->host_converge), sizeof(gfhip::converge_state), hipHostMallocDefault) != hipSuccess) {
        creation_error = "cannot allocate reduction scalars";
        gfhip_destroy_context(ctx.release());                  // frees whatever was allocated
        return nullptr;
    }
    return ctx.release();
}

static void release_kernel(gfhip_kernel *k) {
    for (void *p : k->pack_device) {
        if (p) (void)hipFree(p);
    }
    for (auto &piece : k->pieces) {
        for (void *p : piece.pack_device) {
            if (p) (void)hipFree(p);
        }
        if (piece.module) (void)hipModuleUnload(piece.module);
    }
    for (void *p : k->handover) {
        if (p) (void)hipFree(p);
    }
    for (void *p : k->redo.pack_device) {
        if (p) (void)hipFree(p);
    }
    for (void *p : k->undo) {
        if (p) (void)hipFree(p);
    }
    if (k->redo.module) (void)hipModuleUnload(k->redo.module);
    if (k->flagged) (void)hipFree(k->flagged);
    if (k->redo_list) (void)hipFree(k->redo_list);
    if (k->redo_count) (void)hipFree(k->redo_count);
    for (auto &e : k->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : k->free_events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (k->module) (void)hipModuleUnload(k->module);
}

extern "C" void gfhip_destroy_context(gfhip_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &k : ctx->kernels) release_kernel(k.get());
    for (auto &kv : ctx->buffers) {
        if (kv.second.owned && kv.second.pointer) (void)hipFree(kv.second.pointer);
        if (kv.second.mirror) (void)hipHostFree(kv.second.mirror);
    }
    for (auto &kv : ctx->random_states) {
        if (kv.second.first) (void)hipFree(kv.second.first);
    }
    if (ctx->device_converge) (void)hipFree(ctx->device_converge);
    if (ctx->host_converge) (void)hipHostFree(ctx->host_converge);
    if (ctx->device_scalar) (void)hipFree(ctx->device_scalar);
    if (ctx->device_flags) (void)hipFree(ctx->device_flags);
    if (ctx->host_scalar) (void)hipHostFree(ctx->host_scalar);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

//  How an item is compiled: as one kernel (`pieces` stays empty, `whole` is its lowering), or — items of
//  more than options.segment_nodes records, or every large item when GFHIP_SEGMENTS asks for it — as a
//  sequence of segment kernels (segments.hpp).  Returns the number of hand-over slots.
static uint32_t plan_item(const gfhip::item &it, gfhip::lowered &whole, std::vector<built_piece> &pieces,
                          built_piece *redo = nullptr, bool *has_redo = nullptr) {
    const gfhip::codegen_options opt = gfhip::codegen_options::from_environment();
    size_t count = 1;
    bool automatic = false;
    if (gfhip::can_split(it)) {
//  GFHIP_SEGMENTS=1: the item stays one kernel, but its lanes outside the window are redone by the redo launch
//  instead of an IEEE function compiled into it (whose registers the kernel would have to reserve).
        if (opt.segments >= 1 && it.code.size() >= opt.segments_min_nodes && it.code.size() >= 2*opt.segments) {
            count = opt.segments;
        } else if (opt.segment_nodes && it.code.size() > opt.segment_nodes) {
            count = (it.code.size() + opt.segment_nodes - 1)/opt.segment_nodes;
            automatic = true;
        }
    }
//  GFHIP_ASM=1: the same one-kernel form for the items the assembly body takes (asm_body.hpp).
    bool as_assembly = opt.asm_body && count < 2 && opt.segments == 0 && opt.schedule_for_pressure && gfhip::can_split(it) &&
                       it.code.size() >= opt.asm_min_nodes && opt.division == gfhip::division_mode::shared &&
                       gfhip::asm_body_writer::why_not(it, opt).empty();
    gfhip::item ordered;
    if (as_assembly) {
//  ... if its values fit the register pool and the LDS slots in some emission order (lower() writes the statement of the
//  piece below; here the same writer is asked whether it can).
        std::vector<std::string> directories;
        if (const char *env = std::getenv("GFHIP_CACHE_DIR")) directories.push_back(env);
        directories.push_back(library_directory() + "/kernel_cache");
        ordered = gfhip::schedule_for_assembly(it, opt, directories);
        as_assembly = gfhip::assembly_fits(ordered, opt);
    }
    if (count < 2 && opt.segments != 1 && !as_assembly) {
        whole = gfhip::lower(it, opt);
        return 0;
    }
//  Cut in the pressure-aware emission order where that is affordable (schedule.hpp), else in the order given.
    if (!as_assembly) ordered = opt.schedule_for_pressure ? gfhip::schedule_for_pressure(it) : it;
    gfhip::segmentation plan = gfhip::split_item(ordered, gfhip::choose_cuts(ordered, count));
//  A one-kernel item keeps its name (profiles show gfhip_<name> and gfhip_<name>_redo).
    if (plan.segments.size() == 1) plan.segments[0].piece.name = it.name;
    gfhip::codegen_options piece_options = opt;
//  Very large items are off the hot path: the compiler's division, no checks, no second body.
    if (automatic) piece_options.division = gfhip::division_mode::ieee;
//  The experiment on hot items keeps the shared-reciprocal division and compiles NO IEEE function into the
//  segments (so that they fit two waves per SIMD): lanes outside the window are redone by one more launch.
    const bool with_redo = !automatic && piece_options.division != gfhip::division_mode::ieee;
//  Which values of the item depend on a quotient: a handed-over one keeps that mark in the segments that read it.
    std::vector<bool> after_division(ordered.code.size(), false);
    for (size_t i = 0; i < ordered.code.size(); i++) {
        const gfir_instruction &c = ordered.code[i];
        const uint32_t operands[3] = {c.a, c.b, c.c};
        bool dependent = c.op == GFIR_DIV;
        for (int k = 0; k < gfhip::operand_count(c.op) && !dependent; k++) dependent = after_division[operands[k]];
        after_division[i] = dependent;
    }
    for (size_t p = 0; p < plan.segments.size(); p++) {
        built_piece piece;
        gfhip::piece_info role;
        role.scheduled = opt.schedule_for_pressure;
        if (with_redo) {
            role.role = p + 1 == plan.segments.size() ? gfhip::piece_role::last : gfhip::piece_role::middle;
            for (auto slot : plan.segments[p].output_slot) role.output_handed_over.push_back(slot >= 0);
            for (auto record : plan.segments[p].symbol_record) role.symbol_after_division.push_back(record >= 0 && after_division[record]);
        }
        piece.low = gfhip::lower(plan.segments[p].piece, piece_options, role);
        piece.plan = std::move(plan.segments[p]);
        pieces.push_back(std::move(piece));
    }
    if (has_redo) *has_redo = with_redo;
    if (with_redo && redo) {
        gfhip::piece_info role;
        role.role = gfhip::piece_role::redo;
        gfhip::codegen_options plain = opt;
        plain.division = gfhip::division_mode::ieee;
        plain.waves_per_simd = 0;
        redo->plan.piece = it;
        redo->plan.piece.name = it.name + "_redo";
        redo->low = gfhip::lower(redo->plan.piece, plain, role);
    }
    whole = gfhip::lowered();
    whole.kernel_name = "gfhip_" + it.name;
    whole.block_size = pieces[0].low.block_size;
    whole.input_written.assign(it.symbols.size(), false);
    for (auto &s : it.setters) whole.input_written[s.input] = true;
    for (auto &piece : pieces) whole.hash = whole.hash*1099511628211ull ^ piece.low.hash;
    return plan.slots;
}

extern "C" gfhip_kernel *gfhip_add_kernel(gfhip_context *ctx, const void *gfir, size_t bytes, size_t num_rays) {
    if (!ctx) return nullptr;
    std::unique_ptr<gfhip_kernel> k(new gfhip_kernel);
    k->ctx = ctx;
    k->num_rays = num_rays;
    if (!k->item.parse(gfir, bytes, ctx->error)) {
        return nullptr;
    }
    const uint32_t slots = plan_item(k->item, k->low, k->pieces, &k->redo, &k->has_redo);
    if (!k->pieces.empty()) {
//  Rays per walk of the segment sequence: the hand-over buffers of one chunk stay in the Infinity Cache.
        const gfhip::codegen_options opt = gfhip::codegen_options::from_environment();
        size_t chunk = opt.handover_bytes/(static_cast<size_t> (slots ? slots : 1)*k->item.element_size());
        chunk = chunk/1024*1024;
        if (chunk < 16384) chunk = 16384;
        k->chunk = num_rays < chunk ? num_rays : chunk;
        if (slots == 0) k->chunk = num_rays;         // one piece (the assembly body): nothing is handed over, one walk over all rays
        k->handover.assign(slots, nullptr);
    }
    ctx->kernels.push_back(std::move(k));
    return ctx->kernels.back().get();
}

extern "C" int gfhip_export_piece(const void *gfir, size_t bytes, uint32_t index, void **piece, size_t *piece_bytes) {
    gfhip::item it;
    std::string error;
    if (!piece || !piece_bytes) return 1;
    *piece = nullptr;
    *piece_bytes = 0;
    if (!it.parse(gfir, bytes, error)) {
        creation_error = error;
        return 1;
    }
    gfhip::lowered whole;
    std::vector<built_piece> pieces;
    const uint32_t slots = plan_item(it, whole, pieces);
    if (index >= pieces.size()) return 0;
    const gfhip::segment &plan = pieces[index].plan;
    std::vector<int32_t> head = {static_cast<int32_t> (plan.piece.symbols.size()), static_cast<int32_t> (plan.piece.outputs.size()),
                                 static_cast<int32_t> (slots), static_cast<int32_t> (pieces.size())};
    head.insert(head.end(), plan.symbol_state.begin(), plan.symbol_state.end());
    head.insert(head.end(), plan.symbol_slot.begin(), plan.symbol_slot.end());
    head.insert(head.end(), plan.output_slot.begin(), plan.output_slot.end());
    head.insert(head.end(), plan.output_original.begin(), plan.output_original.end());
    const std::vector<uint8_t> body = plan.piece.serialize();
    *piece_bytes = head.size()*4 + body.size();
    *piece = std::malloc(*piece_bytes);
    std::memcpy(*piece, head.data(), head.size()*4);
    std::memcpy(static_cast<char *> (*piece) + head.size()*4, body.data(), body.size());
    return 0;
}

extern "C" int gfhip_generate_piece_source(const void *gfir, size_t bytes, uint32_t index, char **source, uint64_t *source_hash) {
    gfhip::item it;
    std::string error;
    if (!source) return 1;
    *source = nullptr;
    if (!it.parse(gfir, bytes, error)) {
        creation_error = error;
        return 1;
    }
    gfhip::lowered whole;
    std::vector<built_piece> pieces;
    built_piece redo;
    bool has_redo = false;
    (void)plan_item(it, whole, pieces, &redo, &has_redo);
    const gfhip::lowered *low = nullptr;
    if (pieces.empty()) {
        if (index == 0) low = &whole;
    } else if (index < pieces.size()) {
        low = &pieces[index].low;
    } else if (has_redo && index == pieces.size()) {
        low = &redo.low;
    }
    if (!low) return 0;                                 // past the last piece: *source stays NULL
    if (source_hash) *source_hash = low->hash;
    *source = static_cast<char *> (std::malloc(low->source.size() + 1));
    std::memcpy(*source, low->source.c_str(), low->source.size() + 1);
    return 0;
}

extern "C" char *gfhip_generate_source(const void *gfir, size_t bytes, uint64_t *source_hash) {
    gfhip::item it;
    std::string error;
    if (!it.parse(gfir, bytes, error)) {
        creation_error = error;
        return nullptr;
    }
    gfhip::lowered low;
    std::vector<built_piece> pieces;
    (void)plan_item(it, low, pieces);
//  A segmented item: the texts of its pieces one after the other (each is a translation unit of its
//  own: gfhip_generate_piece_source hands them out one by one).
    for (auto &piece : pieces) low.source += piece.low.source;
    if (source_hash) *source_hash = low.hash;
    char *text = static_cast<char *> (std::malloc(low.source.size() + 1));
    std::memcpy(text, low.source.c_str(), low.source.size() + 1);
    return text;
}

extern "C" void gfhip_free_string(char *text) {
    std::free(text);
}

//  The code object of one lowered kernel text: from the kernel cache (by source hash) or hipRTC.
static int load_code_object(gfhip_context *ctx, const std::string &name, const gfhip::lowered &low,
                            std::vector<char> &code, bool &from_cache) {
    const std::string file = hash_name(low.hash) + ".hsaco";
    std::vector<std::string> directories;
    if (const char *env = std::getenv("GFHIP_CACHE_DIR")) directories.push_back(env);
    directories.push_back(library_directory() + "/kernel_cache");

    from_cache = false;
    for (auto &d : directories) {
        if (read_file(d + "/" + file, code)) {
            from_cache = true;
            return 0;
        }
    }
    if (std::getenv("GFHIP_REQUIRE_CACHE")) {
        return ctx->fail("kernel " + name + " (" + file + ") not in the kernel cache and GFHIP_REQUIRE_CACHE is set");
    }
    hiprtcProgram program;
    if (hiprtcCreateProgram(&program, low.source.c_str(), (name + ".hip").c_str(), 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        return ctx->fail("hiprtcCreateProgram failed");
    }
    const char *options[] = {"-O3", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950"};
    const hiprtcResult result = hiprtcCompileProgram(program, 4, options);
    if (result != HIPRTC_SUCCESS) {
        size_t log_size = 0;
        hiprtcGetProgramLogSize(program, &log_size);
        std::string log(log_size, '\0');
        if (log_size) hiprtcGetProgramLog(program, &log[0]);
        hiprtcDestroyProgram(&program);
        return ctx->fail("hipRTC failed for " + name + ": " + log);
    }
    size_t size = 0;
    hiprtcGetCodeSize(program, &size);
    code.resize(size);
    hiprtcGetCode(program, code.data());
    hiprtcDestroyProgram(&program);
    if (const char *env = std::getenv("GFHIP_CACHE_DIR")) {
        ::mkdir(env, 0755);
        std::ofstream f(std::string(env) + "/" + file, std::ios::binary);
        f.write(code.data(), static_cast<std::streamsize> (code.size()));
    }
    return 0;
}

//  What build_module produces for one lowered item.
struct built_module {
    hipModule_t module = nullptr;
    hipFunction_t function = nullptr, max_function = nullptr, converge_function = nullptr, batch_function = nullptr;
    std::vector<void *> packs;
    int vgprs = 0, lds_static = 0, scratch = 0;
    bool from_cache = false;
    unsigned int grid = 1;
};

//  Build one kernel of `rays` lanes of work per launch.  Module, functions and packs are built into
//  `out` and handed to the caller only when every step has succeeded, so a failed build leaves
//  nothing half-initialised behind (a later gfhip_compile retries).
static int build_module(gfhip_context *ctx, const gfhip::item &item, const gfhip::lowered &low, const size_t rays,
                        built_module &out) {
    std::vector<char> code;
    bool from_cache = false;
    if (load_code_object(ctx, item.name, low, code, from_cache)) return 1;

    hipModule_t module = nullptr;
    hipFunction_t function = nullptr, max_function = nullptr, converge_function = nullptr, batch_function = nullptr;
    std::vector<void *> packs(low.packs.size(), nullptr);
    auto abandon = [&] (const int status) {
        for (void *p : packs) {
            if (p) (void)hipFree(p);
        }
        if (module) (void)hipModuleUnload(module);
        return status;
    };
    if (ctx->check(hipModuleLoadData(&module, code.data()), "hipModuleLoadData")) return abandon(1);
    if (ctx->check(hipModuleGetFunction(&function, module, low.kernel_name.c_str()), "hipModuleGetFunction")) return abandon(1);
    if (low.has_max &&
        ctx->check(hipModuleGetFunction(&max_function, module, (low.kernel_name + "_max").c_str()),
                   "hipModuleGetFunction(max)")) return abandon(1);
    if (low.has_converge &&
        ctx->check(hipModuleGetFunction(&converge_function, module, (low.kernel_name + "_converge").c_str()),
                   "hipModuleGetFunction(converge)")) return abandon(1);
    if (low.batch > 1 &&
        ctx->check(hipModuleGetFunction(&batch_function, module, (low.kernel_name + "_batch").c_str()),
                   "hipModuleGetFunction(batch)")) return abandon(1);
    int vgprs = 0, lds_static = 0, scratch = 0;
    (void)hipFuncGetAttribute(&vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, function);
    (void)hipFuncGetAttribute(&lds_static, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, function);
    (void)hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, function);
    if (low.lds_bytes > 48*1024) {
        for (hipFunction_t f : {function, max_function, converge_function, batch_function}) {
            if (f) (void)hipFuncSetAttribute(reinterpret_cast<const void *> (f), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int> (low.lds_bytes));
        }
    }

//  Pack and upload the tables: [cell][column], padded to the pack stride.
    const size_t esize = item.element_size();
    const size_t parts = item.is_complex() ? 2 : 1;
    const bool wide = item.base_is_f64();
    for (size_t p = 0; p < low.packs.size(); p++) {
        const gfhip::pack &pk = low.packs[p];
        const size_t cells = pk.cells();
        std::vector<unsigned char> host(pk.elements()*esize, 0);
        for (size_t column = 0; column < pk.tables.size(); column++) {
            const gfhip::table &t = item.tables[pk.tables[column]];
            for (size_t cell = 0; cell < cells; cell++) {
                for (size_t part = 0; part < parts; part++) {
                    const size_t at = (cell*pk.stride + column)*parts + part;
                    if (wide) {
                        reinterpret_cast<double *> (host.data())[at] = t.data[cell*parts + part];
                    } else {
                        reinterpret_cast<float *> (host.data())[at] = static_cast<float> (t.data[cell*parts + part]);
                    }
                }
            }
        }
        if (ctx->check(hipMalloc(&packs[p], host.size() ? host.size() : 8), "hipMalloc(pack)")) return abandon(1);
        if (ctx->check(hipMemcpy(packs[p], host.data(), host.size(), hipMemcpyHostToDevice), "hipMemcpy(pack)")) return abandon(1);
    }

//  Launch geometry: one lane per ray; the kernel grid-strides, so cap the grid
//  at a few waves of workgroups per CU.
    const size_t block = low.block_size;
    size_t want = (rays + block - 1)/block;
    if (want < 1) want = 1;
//  Persistent-style grid: a few workgroups per resident slot, the kernel grid-strides.
//  Measured (1e7-particle fp64 push): exact grid 0.276 ms, 64 per CU 0.245, 16 per CU 0.235.
//  Register-bound items that fit one workgroup per CU (the RK4 kernel: 507 registers, one wave
//  per SIMD) run best with exactly one workgroup per CU: 0.279 vs 0.298 ms per step at 1e6
//  rays (the coefficient packs are staged into LDS once per workgroup instead of 15 times).
    int resident = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&resident, function, static_cast<int> (block),
                                                           low.lds_bytes) != hipSuccess || resident < 1) {
        resident = 1;
    }
    size_t cap = static_cast<size_t> (ctx->num_cus)*static_cast<size_t> (resident)*(resident == 1 ? 1 : 4);
//  The assembly body (two workgroups resident per CU, VALU busy 98 % of the time its waves are resident): the finer the
//  grid the better the CUs finish together — 2 per CU 1.90 ms per RK4 step at 1e7 rays, 8: 1.79, 16..48: 1.76, 64: 1.75,
//  one workgroup per tile (153): 1.78 (profiles/r03_asm_grid.jsonl).
    if (low.assembly) cap = static_cast<size_t> (ctx->num_cus)*64u;
    if (const char *env = std::getenv("GFHIP_GRID_PER_CU")) {
        cap = static_cast<size_t> (ctx->num_cus)*static_cast<size_t> (std::atoi(env) > 0 ? std::atoi(env) : 1);
    }

    out.module = module;
    out.function = function;
    out.max_function = max_function;
    out.converge_function = converge_function;
    out.batch_function = batch_function;
    out.packs = packs;
    out.vgprs = vgprs;
    out.lds_static = lds_static;
    out.scratch = scratch;
    out.from_cache = from_cache;
    out.grid = static_cast<unsigned int> (want < cap ? want : cap);
//  A kernel that draws random numbers: lane t owns MT19937 state t of 1024 and serves elements
//  t, t + 1024, ... in order (cuda_context.hpp:509-522 launches one 1024-thread block per 1024
//  elements, one after the other).
    if (item.has_random() && out.grid*block > 1024) out.grid = static_cast<unsigned int> (1024/block);
    return 0;
}

static int build_kernel(gfhip_context *ctx, gfhip_kernel *k) {
    if (!k->pieces.empty()) {
//  A segmented item: every piece is a kernel of its own over one chunk of rays; one device array
//  per hand-over slot.  Everything is committed only when every piece has been built.
        std::vector<built_module> modules(k->pieces.size());
        std::vector<void *> handover(k->handover.size(), nullptr);
        auto abandon = [&] (const int status) {
            for (auto &m : modules) {
                for (void *p : m.packs) {
                    if (p) (void)hipFree(p);
                }
                if (m.module) (void)hipModuleUnload(m.module);
            }
            for (void *p : handover) {
                if (p) (void)hipFree(p);
            }
            return status;
        };
        for (size_t p = 0; p < k->pieces.size(); p++) {
            if (build_module(ctx, k->pieces[p].plan.piece, k->pieces[p].low, k->chunk, modules[p])) return abandon(1);
        }
        const size_t bytes = k->chunk*k->item.element_size();
        for (auto &slot : handover) {
            if (ctx->check(hipMalloc(&slot, bytes ? bytes : 8), "hipMalloc(hand-over)")) return abandon(1);
        }
        if (k->has_redo) {
            built_module redo;
            if (build_module(ctx, k->redo.plan.piece, k->redo.low, 64*256, redo)) return abandon(1);
            const size_t rays = k->num_rays ? k->num_rays : 1;
            if (ctx->check(hipMalloc(reinterpret_cast<void **> (&k->flagged), rays), "hipMalloc(flagged)") ||
                ctx->check(hipMemset(k->flagged, 0, rays), "hipMemset(flagged)") ||
                ctx->check(hipMalloc(reinterpret_cast<void **> (&k->redo_list), rays*sizeof(unsigned int)), "hipMalloc(redo list)") ||
                ctx->check(hipMalloc(reinterpret_cast<void **> (&k->redo_count), 2*sizeof(unsigned int)), "hipMalloc(redo count)") ||
                ctx->check(hipMemset(k->redo_count, 0, 2*sizeof(unsigned int)), "hipMemset(redo count)")) {
                for (void *p : redo.packs) {
                    if (p) (void)hipFree(p);
                }
                (void)hipModuleUnload(redo.module);
                return abandon(1);
            }
            k->redo.module = redo.module;
            k->redo.function = redo.function;
            k->redo.pack_device = redo.packs;
            k->redo.grid = redo.grid;
            k->redo.vgprs = redo.vgprs;
            k->redo.scratch = redo.scratch;
            k->redo.from_cache = redo.from_cache;
        }
        for (size_t p = 0; p < k->pieces.size(); p++) {
            built_piece &piece = k->pieces[p];
            piece.module = modules[p].module;
            piece.function = modules[p].function;
            piece.pack_device = modules[p].packs;
            piece.grid = modules[p].grid;
            piece.vgprs = modules[p].vgprs;
            piece.lds_static = modules[p].lds_static;
            piece.scratch = modules[p].scratch;
            piece.from_cache = modules[p].from_cache;
            k->vgprs = std::max(k->vgprs, piece.vgprs);
            k->scratch = std::max(k->scratch, piece.scratch);
            k->lds_static = std::max(k->lds_static, piece.lds_static);
        }
        k->handover = handover;
        k->from_cache = true;
        for (auto &piece : k->pieces) k->from_cache = k->from_cache && piece.from_cache;
        k->grid = k->pieces[0].grid;
        k->built = true;
        return 0;
    }
    built_module built;
    if (build_module(ctx, k->item, k->low, k->num_rays, built)) return 1;
    k->module = built.module;
    k->function = built.function;
    k->max_function = built.max_function;
    k->converge_function = built.converge_function;
    k->batch_function = built.batch_function;
    k->pack_device = built.packs;
    k->vgprs = built.vgprs;
    k->lds_static = built.lds_static;
    k->scratch = built.scratch;
    k->from_cache = built.from_cache;
    k->grid = built.grid;
    k->built = true;
    return 0;
}

//  Passes that ran ahead of a caller iterating on gfhip_run_max are taken back before anything else looks at the state
//  (defined with the batch launches below).
static int settle(gfhip_context *ctx);

extern "C" int gfhip_compile(gfhip_context *ctx) {
    if (!ctx) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    for (auto &k : ctx->kernels) {
        if (!k->built) {
            if (build_kernel(ctx, k.get())) return 1;
        }
    }
    return 0;
}

static int ensure_buffer(gfhip_context *ctx, const uint64_t key, const size_t count, const uint32_t dtype,
                         const void *init, size_t init_count = ~static_cast<size_t> (0)) {
    auto found = ctx->buffers.find(key);
    if (found == ctx->buffers.end()) {
        buffer b;
        b.count = count;
        b.dtype = dtype;
        const size_t esize = element_bytes(dtype);
        const size_t bytes = count*esize;
        if (init_count > count) init_count = count;
        GFHIP_TRY(ctx, hipMalloc(&b.pointer, bytes ? bytes : 8), "hipMalloc(buffer)");
        if (!init || init_count < count) {
            GFHIP_TRY(ctx, hipMemset(b.pointer, 0, bytes), "hipMemset(buffer)");
        }
        if (init && init_count) {
            const hipError_t status = hipMemcpy(b.pointer, init, init_count*esize, hipMemcpyHostToDevice);
            if (status != hipSuccess) {
                (void)hipFree(b.pointer);
                return ctx->check(status, "hipMemcpy(init)");
            }
        }
        ctx->buffers[key] = b;
        return 0;
    }
    if (found->second.count < count) {
        return ctx->fail("buffer is smaller than the kernel's ensemble size");
    }
    if (found->second.dtype != dtype) {
        return ctx->fail("buffer element type does not match the kernel");
    }
    return 0;
}

extern "C" int gfhip_create_kernel_call(gfhip_kernel *k, const uint64_t *input_keys,
                                        const void *const *input_init, const size_t *input_counts,
                                        const uint64_t *output_keys) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    const size_t ni = k->item.symbols.size(), no = k->item.outputs.size();
    k->input_keys.assign(input_keys, input_keys + ni);
    k->output_keys.assign(output_keys, output_keys + no);
    for (size_t i = 0; i < ni; i++) {
//  An input that index nodes read (index_1D/2D_node) is a buffer of its own length.
        const size_t indexed = k->item.indexed_length(static_cast<uint32_t> (i));
        const size_t given = input_counts ? input_counts[i] : k->num_rays;
        size_t count = k->num_rays;
        if (indexed > count) count = indexed;
        if (input_init && input_init[i] && given > count) count = given;
        if (input_init && input_init[i] && given < indexed) {
            return ctx->fail("initial values of an input that index nodes read are shorter than the indexed length");
        }
        if (ensure_buffer(ctx, input_keys[i], count, k->item.dtype, input_init ? input_init[i] : nullptr, given)) return 1;
    }
    for (size_t o = 0; o < no; o++) {
        if (ensure_buffer(ctx, output_keys[o], k->num_rays, k->item.dtype, nullptr)) return 1;
    }
//  The kernel's pointers are __restrict__: written buffers must be distinct.
    for (size_t o = 0; o < no; o++) {
        for (size_t i = 0; i < ni; i++) {
            if (output_keys[o] == input_keys[i]) return ctx->fail("an output buffer aliases an input buffer");
        }
        for (size_t p = o + 1; p < no; p++) {
            if (output_keys[o] == output_keys[p]) return ctx->fail("two outputs share one buffer");
        }
    }
    for (size_t i = 0; i < ni; i++) {
        for (size_t j = i + 1; j < ni; j++) {
            if (input_keys[i] == input_keys[j]) return ctx->fail("two inputs share one buffer");
        }
    }
    k->bound = true;
    return 0;
}

//  Launch `<name>` (reduce == nullptr) or `<name>_max` (the max of the last output is folded into
//  *reduce; a non-null `stop` word that reads non-zero makes the launch return at once).
//  A segmented item (segments.hpp): `steps` passes, each a walk over the ensemble in chunks, each chunk
//  through the sequence of segment kernels.  A piece's symbols are state arrays (offset to the chunk) or
//  hand-over slots; its outputs are slots or, in the last piece, the item's outputs.  Only the last
//  piece stores state, so a chunk's pieces all read the state of the beginning of the pass.
static int launch_pieces(gfhip_kernel *k, const uint32_t steps) {
    gfhip_context *ctx = k->ctx;
    const size_t esize = k->item.element_size();
    for (uint32_t step = 0; step < steps; step++) {
        for (size_t first = 0; first < k->num_rays; first += k->chunk) {
            unsigned long long n = std::min(k->chunk, k->num_rays - first);
            for (auto &piece : k->pieces) {
                std::vector<void *> pointers;
                for (size_t i = 0; i < piece.plan.piece.symbols.size(); i++) {
                    if (piece.plan.symbol_state[i] >= 0) {
                        pointers.push_back(static_cast<char *> (ctx->buffers[k->input_keys[piece.plan.symbol_state[i]]].pointer) + first*esize);
                    } else {
                        pointers.push_back(k->handover[piece.plan.symbol_slot[i]]);
                    }
                }
                for (size_t o = 0; o < piece.plan.piece.outputs.size(); o++) {
                    if (piece.plan.output_slot[o] >= 0) {
                        pointers.push_back(k->handover[piece.plan.output_slot[o]]);
                    } else {
                        pointers.push_back(static_cast<char *> (ctx->buffers[k->output_keys[piece.plan.output_original[o]]].pointer) + first*esize);
                    }
                }
                for (void *p : piece.pack_device) pointers.push_back(p);
                pointers.push_back(ctx->device_flags);
                unsigned int one = 1;
                unsigned int first_ray = static_cast<unsigned int> (first);
                unsigned char *flagged = k->flagged ? k->flagged + first : nullptr;
                std::vector<void *> params;
                for (auto &p : pointers) params.push_back(&p);
                params.push_back(&n);
                if (k->has_redo) {
                    params.push_back(&flagged);
                    if (&piece == &k->pieces.back()) {
                        params.push_back(&k->redo_list);
                        params.push_back(&k->redo_count);
                        params.push_back(&first_ray);
                    }
                }
                params.push_back(&one);
                const size_t want = (n + piece.low.block_size - 1)/piece.low.block_size;
                const unsigned int grid = static_cast<unsigned int> (want < piece.grid ? want : piece.grid);
                GFHIP_TRY(ctx, hipModuleLaunchKernel(piece.function, grid, 1, 1, piece.low.block_size, 1, 1,
                                                     static_cast<unsigned int> (piece.low.lds_bytes), ctx->stream,
                                                     params.data(), nullptr), "hipModuleLaunchKernel(segment)");
            }
        }
        if (k->has_redo) {
//  The lanes the segments left alone: the whole item with the compiler's division, from the untouched state.
            std::vector<void *> pointers;
            for (auto key : k->input_keys) pointers.push_back(ctx->buffers[key].pointer);
            for (auto key : k->output_keys) pointers.push_back(ctx->buffers[key].pointer);
            for (void *p : k->redo.pack_device) pointers.push_back(p);
            pointers.push_back(ctx->device_flags);
            unsigned long long n = k->num_rays;
            unsigned int one = 1;
            std::vector<void *> params;
            for (auto &p : pointers) params.push_back(&p);
            params.push_back(&n);
            params.push_back(&k->flagged);
            params.push_back(&k->redo_list);
            params.push_back(&k->redo_count);
            params.push_back(&one);
//  One workgroup per CU: an empty list costs the launch either way (5 us), a long one (the O-mode step on the CLI beam
//  sends a third of its lanes here) is walked by the whole chip.
            GFHIP_TRY(ctx, hipModuleLaunchKernel(k->redo.function, ctx->num_cus, 1, 1, k->redo.low.block_size, 1, 1,
                                                 static_cast<unsigned int> (k->redo.low.lds_bytes), ctx->stream,
                                                 params.data(), nullptr), "hipModuleLaunchKernel(redo)");
//  (the redo kernel leaves the count — and its arrival counter, the second word — at zero)
        }
    }
    return 0;
}

static int launch(gfhip_kernel *k, const uint32_t steps, unsigned long long *reduce = nullptr,
                  const unsigned int *stop = nullptr) {
    gfhip_context *ctx = k->ctx;
    if (!k->built) return ctx->fail("kernel has not been compiled (gfhip_compile)");
    if (!k->bound) return ctx->fail("kernel arguments are not bound (gfhip_create_kernel_call)");
    if (k->num_rays == 0 || steps == 0) return 0;
    if (reduce && !k->max_function) return ctx->fail("item has no in-launch max reduction");
    if (!k->pieces.empty()) {
        std::pair<hipEvent_t, hipEvent_t> ev;
        const bool timed = ctx->timing && (k->launch_count++ % ctx->timing) == 0;
        if (timed) {
            if (!k->free_events.empty()) {
                ev = k->free_events.back();
                k->free_events.pop_back();
            } else {
                GFHIP_TRY(ctx, hipEventCreate(&ev.first), "hipEventCreate");
                GFHIP_TRY(ctx, hipEventCreate(&ev.second), "hipEventCreate");
            }
            GFHIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream), "hipEventRecord");
        }
        if (launch_pieces(k, steps)) return 1;
        if (timed) {
            GFHIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream), "hipEventRecord");
            k->events.push_back(ev);
        }
        return 0;
    }

    std::vector<void *> pointers;
    for (auto key : k->input_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (auto key : k->output_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (void *p : k->pack_device) pointers.push_back(p);
    if (k->item.has_random()) {
        if (!k->random_states) return ctx->fail("the item draws random numbers but no random state is bound (gfhip_set_random_state)");
        pointers.push_back(k->random_states);
    }
    pointers.push_back(ctx->device_flags);
    unsigned long long n = k->num_rays;
    unsigned int step_count = steps;
    std::vector<void *> params;
    for (auto &p : pointers) params.push_back(&p);
    params.push_back(&n);
    params.push_back(&step_count);
    if (reduce) {
        params.push_back(&reduce);
        params.push_back(&stop);
    }

    std::pair<hipEvent_t, hipEvent_t> ev;
    const bool timed = ctx->timing && (k->launch_count++ % ctx->timing) == 0;
    if (timed) {
        if (!k->free_events.empty()) {
            ev = k->free_events.back();
            k->free_events.pop_back();
        } else {
            GFHIP_TRY(ctx, hipEventCreate(&ev.first), "hipEventCreate");
            GFHIP_TRY(ctx, hipEventCreate(&ev.second), "hipEventCreate");
        }
        GFHIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream), "hipEventRecord");
    }
    GFHIP_TRY(ctx, hipModuleLaunchKernel(reduce ? k->max_function : k->function, k->grid, 1, 1, k->low.block_size, 1, 1,
                                         static_cast<unsigned int> (k->low.lds_bytes), ctx->stream,
                                         params.data(), nullptr), "hipModuleLaunchKernel");
    if (timed) {
        GFHIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream), "hipEventRecord");
        k->events.push_back(ev);
    }
    return 0;
}

extern "C" int gfhip_run(gfhip_kernel *k, uint32_t steps) {
    if (!k) return 1;
    GFHIP_TRY(k->ctx, hipSetDevice(k->ctx->device), "hipSetDevice");
    if (settle(k->ctx)) return 1;
    return launch(k, steps);
}

static double decode_ordered(const unsigned long long key, const bool f64) {
    if (f64) {
        const unsigned long long bits = (key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key;
        double v;
        std::memcpy(&v, &bits, 8);
        return v;
    }
    const unsigned int k32 = static_cast<unsigned int> (key);
    const unsigned int bits = (k32 >> 31) ? (k32 & 0x7FFFFFFFu) : ~k32;
    float v;
    std::memcpy(&v, &bits, 4);
    return static_cast<double> (v);
}

//  One pass + the max of its last output left in ctx->device_scalar (ordered image), no sync:
//  inside the launch for items that have `<name>_max`, else the separate reduction kernel.
static int enqueue_pass_with_max(gfhip_kernel *k, const unsigned int *stop) {
    gfhip_context *ctx = k->ctx;
    if (k->max_function) {
        return launch(k, 1, ctx->device_scalar, stop);
    }
    if (launch(k, 1)) return 1;
    const buffer &b = ctx->buffers[k->output_keys.back()];
    gfhip::launch_max_reduce(b.pointer, k->num_rays, k->item.dtype == GFIR_F64, ctx->device_scalar,
                             ctx->num_cus, ctx->stream);
    GFHIP_TRY(ctx, hipGetLastError(), "max_reduce launch");
    return 0;
}

//  Complex items: run, then the element of largest modulus of the last output, as
//  cpu_context.hpp:314-318 selects it (std::max_element on std::abs, the first of equals).
extern "C" int gfhip_run_max_complex(gfhip_kernel *k, double *value) {
    if (!k || !value) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (k->output_keys.empty()) return ctx->fail("converge item has no output to reduce");
    if (!k->item.is_complex()) {
        value[1] = 0.0;
        return gfhip_run_max(k, value);
    }
    if (settle(ctx)) return 1;
    if (launch(k, 1)) return 1;
    value[0] = value[1] = 0.0;
    if (k->num_rays == 0) return 0;
    const buffer &b = ctx->buffers[k->output_keys.back()];
    const bool wide = k->item.base_is_f64();
    gfhip::launch_max_modulus(b.pointer, k->num_rays, wide, ctx->device_converge, ctx->stream);
    GFHIP_TRY(ctx, hipGetLastError(), "max_modulus launch");
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_converge, ctx->device_converge, 16, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    if (wide) {
        std::memcpy(value, ctx->host_converge, 16);
    } else {
        float narrow[2];
        std::memcpy(narrow, ctx->host_converge, 8);
        value[0] = narrow[0];
        value[1] = narrow[1];
    }
    return 0;
}

extern "C" int gfhip_reduce_max(gfhip_context *ctx, uint64_t key, double *value) {
    if (!ctx || !value) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    auto found = ctx->buffers.find(key);
    if (found == ctx->buffers.end()) return ctx->fail("unknown buffer key");
    const buffer &b = found->second;
    const bool wide = b.dtype == GFIR_F64 || b.dtype == GFIR_C64;
    value[0] = value[1] = 0.0;
    if (b.dtype == GFIR_C32 || b.dtype == GFIR_C64) {
        if (b.count == 0) return 0;
        gfhip::launch_max_modulus(b.pointer, b.count, wide, ctx->device_converge, ctx->stream);
        GFHIP_TRY(ctx, hipGetLastError(), "max_modulus launch");
        GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_converge, ctx->device_converge, 16, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
        GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        if (wide) {
            std::memcpy(value, ctx->host_converge, 16);
        } else {
            float narrow[2];
            std::memcpy(narrow, ctx->host_converge, 8);
            value[0] = narrow[0];
            value[1] = narrow[1];
        }
        return 0;
    }
    if (b.count == 0) {
        value[0] = -std::numeric_limits<double>::infinity();
        return 0;
    }
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    gfhip::launch_max_reduce(b.pointer, b.count, wide, ctx->device_scalar, ctx->num_cus, ctx->stream);
    GFHIP_TRY(ctx, hipGetLastError(), "max_reduce launch");
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_scalar, ctx->device_scalar, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    value[0] = decode_ordered(*ctx->host_scalar, wide);
    return 0;
}

static int run_max_ahead(gfhip_kernel *k, double *max_value, bool &answered);

extern "C" int gfhip_run_max(gfhip_kernel *k, double *max_value) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (k->output_keys.empty()) return ctx->fail("converge item has no output to reduce");
    if (k->item.is_complex()) return ctx->fail("complex item: use gfhip_run_max_complex");
    bool answered = false;
    if (run_max_ahead(k, max_value, answered)) return 1;
    if (answered) return 0;
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    if (enqueue_pass_with_max(k, nullptr)) return 1;
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_scalar, ctx->device_scalar, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    *max_value = k->num_rays ? decode_ordered(*ctx->host_scalar, k->item.dtype == GFIR_F64)
                             : -std::numeric_limits<double>::infinity();
    return 0;
}

//  workflow::converge_item::run, workflow.hpp:179-205, in the item's own type, one host
//  synchronisation per pass: the form for items without `<name>_max` and for empty ensembles.
template<typename T>
static int converge_loop(gfhip_kernel *k, const double tolerance_, const size_t max_iterations,
                         size_t *iterations_out, double *last_max) {
    const T tolerance = static_cast<T> (tolerance_);
    size_t iterations = 0;
    double value;
    if (gfhip_run_max(k, &value)) return 1;
    T max_residual = static_cast<T> (value);
    T last = std::numeric_limits<T>::max();
    T off_last = std::numeric_limits<T>::max();
    while (std::abs(max_residual) > std::abs(tolerance)            &&
           std::abs(last - max_residual) > std::abs(tolerance)     &&
           std::abs(off_last - max_residual) > std::abs(tolerance) &&
           iterations++ < max_iterations) {
        last = max_residual;
        if (!(iterations%2)) {
            off_last = max_residual;
        }
        if (gfhip_run_max(k, &value)) return 1;
        max_residual = static_cast<T> (value);
    }
    if (iterations_out) *iterations_out = iterations;
    if (last_max) *last_max = static_cast<double> (max_residual);
    return 0;
}

//  ... for complex items (max = the element of largest modulus; the loop compares moduli,
//  workflow.hpp:183-186 with T = std::complex).
template<typename B>
static int converge_loop_complex(gfhip_kernel *k, const double tolerance_, const size_t max_iterations,
                                 size_t *iterations_out, double *last_max) {
    typedef std::complex<B> T;
    const T tolerance(static_cast<B> (tolerance_), 0);
    size_t iterations = 0;
    double value[2];
    if (gfhip_run_max_complex(k, value)) return 1;
    T max_residual(static_cast<B> (value[0]), static_cast<B> (value[1]));
    T last = std::numeric_limits<T>::max();            // std::numeric_limits<std::complex> is the unspecialised one: T()
    T off_last = std::numeric_limits<T>::max();
    while (std::abs(max_residual) > std::abs(tolerance)            &&
           std::abs(last - max_residual) > std::abs(tolerance)     &&
           std::abs(off_last - max_residual) > std::abs(tolerance) &&
           iterations++ < max_iterations) {
        last = max_residual;
        if (!(iterations%2)) {
            off_last = max_residual;
        }
        if (gfhip_run_max_complex(k, value)) return 1;
        max_residual = T(static_cast<B> (value[0]), static_cast<B> (value[1]));
    }
    if (iterations_out) *iterations_out = iterations;
    if (last_max) *last_max = static_cast<double> (std::abs(max_residual));
    return 0;
}

//  The same loop with its test on the device (reduce.hip: converge_decide_kernel): passes are
//  enqueued ahead of the host in growing batches, each followed by the one-thread test; once
//  the test has come out false the passes still queued return at once (`stop`), so exactly the
//  passes of the host loop run, with the same iteration count — and the host synchronises once
//  per batch (twice for the benchmark's 25 passes) instead of once per pass.
static int converge_on_device(gfhip_kernel *k, const double tolerance, const size_t max_iterations,
                              size_t *iterations_out, double *last_max) {
    gfhip_context *ctx = k->ctx;
    const bool f64 = k->item.dtype == GFIR_F64;
    gfhip::converge_state &host = *ctx->host_converge;
    host = gfhip::converge_state();
    host.last = host.off_last = f64 ? std::numeric_limits<double>::max()
                                    : static_cast<double> (std::numeric_limits<float>::max());
    host.tolerance = tolerance;
    host.limit = max_iterations;
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->device_converge, &host, sizeof(host), hipMemcpyHostToDevice, ctx->stream),
              "hipMemcpyAsync(converge state)");
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");     // `host` is written again below
    const unsigned int *stop = &ctx->device_converge->done;
    const uint64_t first_launch = k->launch_count;
    const size_t events_before = k->events.size();
    size_t batch = 16;
    for (;;) {
        for (size_t p = 0; p < batch; p++) {
            if (enqueue_pass_with_max(k, stop)) return 1;
            gfhip::launch_converge_decide(f64, ctx->device_scalar, ctx->device_converge, ctx->stream);
            GFHIP_TRY(ctx, hipGetLastError(), "converge_decide launch");
        }
        GFHIP_TRY(ctx, hipMemcpyAsync(&host, ctx->device_converge, sizeof(host), hipMemcpyDeviceToHost, ctx->stream),
                  "hipMemcpyAsync(converge state)");
        GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        if (host.done) break;
        if (batch < 64) batch *= 2;
    }
//  Launch timing: keep the event pairs of the passes that ran (the first `passes` launches of the
//  loop); the queued launches that returned at once are not launches of the item's work.
    if (ctx->timing) {
        size_t ran = 0;
        for (uint64_t l = first_launch; l < first_launch + host.passes; l++) {
            if (l % ctx->timing == 0) ran++;
        }
        while (k->events.size() > events_before + ran) {
            k->free_events.push_back(k->events.back());
            k->events.pop_back();
        }
    }
    if (iterations_out) *iterations_out = static_cast<size_t> (host.iterations);
    if (last_max) *last_max = host.max_residual;
    return 0;
}

//  One launch of `<name>_batch`: `passes` passes on state kept in registers, the max of each pass folded
//  into reduce[pass]; the setter targets as they were before the launch are saved in the undo arrays.
static int launch_batch(gfhip_kernel *k, const unsigned int passes, const unsigned int *stop) {
    gfhip_context *ctx = k->ctx;
    std::vector<void *> pointers;
    for (auto key : k->input_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (auto key : k->output_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (void *p : k->pack_device) pointers.push_back(p);
    pointers.push_back(ctx->device_flags);
    unsigned long long n = k->num_rays;
    unsigned int count = passes;
    std::vector<void *> params;
    for (auto &p : pointers) params.push_back(&p);
    params.push_back(&n);
    params.push_back(&count);
    params.push_back(&ctx->device_scalar);
    params.push_back(&stop);
    for (auto &u : k->undo) params.push_back(&u);
    std::pair<hipEvent_t, hipEvent_t> ev;
    const bool timed = ctx->timing && (k->launch_count++ % ctx->timing) == 0;
    if (timed) {
        if (!k->free_events.empty()) {
            ev = k->free_events.back();
            k->free_events.pop_back();
        } else {
            GFHIP_TRY(ctx, hipEventCreate(&ev.first), "hipEventCreate");
            GFHIP_TRY(ctx, hipEventCreate(&ev.second), "hipEventCreate");
        }
        GFHIP_TRY(ctx, hipEventRecord(ev.first, ctx->stream), "hipEventRecord");
    }
    GFHIP_TRY(ctx, hipModuleLaunchKernel(k->batch_function, k->grid, 1, 1, k->low.block_size, 1, 1,
                                         static_cast<unsigned int> (k->low.lds_bytes), ctx->stream,
                                         params.data(), nullptr), "hipModuleLaunchKernel(batch)");
    if (timed) {
        GFHIP_TRY(ctx, hipEventRecord(ev.second, ctx->stream), "hipEventRecord");
        k->events.push_back(ev);
    }
    return 0;
}

//  gfhip_run_max called again and again with nothing in between is the reference's converge loop seen from below
//  (workflow.hpp:179-205: the host tests every max; hip_context's create_max_call forwards each call here).  From the
//  second call of such a streak on, a `<name>_batch` launch runs the pass that is asked for AND the next ones, each with
//  its own max; the following calls are answered from those maxes without a launch or a synchronisation.  Whatever
//  entry point comes next (settle) takes back the passes nobody asked for: the state of the beginning of the batch is
//  in the undo arrays, the passes that were asked for run again.  25 passes of the benchmark's Newton solve: 10 launches
//  and host synchronisations instead of 25.  GFHIP_RUN_AHEAD=0 turns it off.
static int settle(gfhip_context *ctx) {
    ctx->max_streak = nullptr;
    gfhip_kernel *k = ctx->running_ahead;
    if (!k) return 0;
    ctx->running_ahead = nullptr;
    const unsigned int asked = k->ahead_taken, ran = static_cast<unsigned int> (k->ahead.size());
    k->ahead.clear();
    k->ahead_taken = 0;
    if (asked == ran) return 0;
    const size_t esize = k->item.element_size();
    for (size_t s = 0; s < k->item.setters.size(); s++) {
        void *target = ctx->buffers[k->input_keys[k->item.setters[s].input]].pointer;
        GFHIP_TRY(ctx, hipMemcpyAsync(target, k->undo[s], k->num_rays*esize, hipMemcpyDeviceToDevice, ctx->stream),
                  "hipMemcpyAsync(undo)");
    }
    return launch_batch(k, asked, nullptr);
}

static int run_max_ahead(gfhip_kernel *k, double *max_value, bool &answered) {
    gfhip_context *ctx = k->ctx;
    answered = false;
    static const bool enabled = !(std::getenv("GFHIP_RUN_AHEAD") && std::string(std::getenv("GFHIP_RUN_AHEAD")) == "0");
    if (ctx->running_ahead == k && k->ahead_taken < k->ahead.size()) {
        *max_value = k->ahead[k->ahead_taken++];
        answered = true;
        return 0;
    }
    const bool streak = ctx->max_streak == k;
    if (ctx->running_ahead == k) {
//  every pass of the last batch was asked for: nothing to take back
        ctx->running_ahead = nullptr;
        k->ahead.clear();
        k->ahead_taken = 0;
    } else if (settle(ctx)) {
        return 1;
    }
    ctx->max_streak = k;
    if (!enabled || !streak || !k->batch_function || k->num_rays == 0 || k->low.batch < 2) return 0;
    if (k->undo.empty()) {
        const size_t esize = k->item.element_size();
        for (size_t s = 0; s < k->item.setters.size(); s++) {
            void *p = nullptr;
            GFHIP_TRY(ctx, hipMalloc(&p, k->num_rays*esize), "hipMalloc(undo)");
            k->undo.push_back(p);
        }
    }
    const unsigned int batch = std::min(k->low.batch, 8u);
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, 8*sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    if (launch_batch(k, batch, nullptr)) return 1;
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_scalar, ctx->device_scalar, batch*sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                  ctx->stream), "hipMemcpyAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    k->ahead.clear();
    for (unsigned int b = 0; b < batch; b++) k->ahead.push_back(decode_ordered(ctx->host_scalar[b], k->item.dtype == GFIR_F64));
    k->ahead_taken = 1;
    ctx->running_ahead = k;
    *max_value = k->ahead[0];
    answered = true;
    return 0;
}

//  The converge loop with several passes per launch (`<name>_batch`, codegen.hpp).  A pass of this loop
//  only feeds the next pass of the same ray and the max the loop's test looks at, so a launch may run a
//  few passes on state kept in registers — each pass leaving its own max — and the test (on the device,
//  reduce.hip: converge_decide_batch_kernel) is applied to those maxima in order afterwards.  If the loop
//  turns out to have ended before the last pass of a batch, the state of the beginning of that batch is
//  restored from the undo arrays and the batch is redone with exactly the passes the loop ran: the same
//  passes, iteration count, state and output as one launch per pass, at a third of the sweeps over the
//  state.  Batches are queued ahead of the host like the single passes of converge_on_device.
static int converge_batched(gfhip_kernel *k, const double tolerance, const size_t max_iterations,
                            size_t *iterations_out, double *last_max) {
    gfhip_context *ctx = k->ctx;
    const bool f64 = k->item.dtype == GFIR_F64;
    const size_t esize = k->item.element_size();
    const unsigned int batch = k->low.batch;
    if (k->undo.empty()) {
        for (size_t s = 0; s < k->item.setters.size(); s++) {
            void *p = nullptr;
            GFHIP_TRY(ctx, hipMalloc(&p, k->num_rays*esize), "hipMalloc(undo)");
            k->undo.push_back(p);
        }
    }
    gfhip::converge_state &host = *ctx->host_converge;
    host = gfhip::converge_state();
    host.last = host.off_last = f64 ? std::numeric_limits<double>::max()
                                    : static_cast<double> (std::numeric_limits<float>::max());
    host.tolerance = tolerance;
    host.limit = max_iterations;
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->device_converge, &host, sizeof(host), hipMemcpyHostToDevice, ctx->stream),
              "hipMemcpyAsync(converge state)");
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, 8*sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");     // `host` is written again below
    const unsigned int *stop = &ctx->device_converge->done;
    const uint64_t first_launch = k->launch_count;
    const size_t events_before = k->events.size();
    size_t queued = 6;                                     // batches per host synchronisation, growing
    for (;;) {
        for (size_t b = 0; b < queued; b++) {
            if (launch_batch(k, batch, stop)) return 1;
            gfhip::launch_converge_decide_batch(f64, ctx->device_scalar, ctx->device_converge, batch, ctx->stream);
            GFHIP_TRY(ctx, hipGetLastError(), "converge_decide launch");
        }
        GFHIP_TRY(ctx, hipMemcpyAsync(&host, ctx->device_converge, sizeof(host), hipMemcpyDeviceToHost, ctx->stream),
                  "hipMemcpyAsync(converge state)");
        GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        if (host.done) break;
        if (queued < 24) queued *= 2;
    }
//  Launch timing: keep the event pairs of the batches that ran; the queued launches that returned at once
//  are not launches of the item's work.
    if (ctx->timing) {
        const uint64_t batches = (static_cast<uint64_t> (host.passes) + host.extra + batch - 1)/batch;
        size_t ran = 0;
        for (uint64_t l = first_launch; l < first_launch + batches; l++) {
            if (l % ctx->timing == 0) ran++;
        }
        while (k->events.size() > events_before + ran) {
            k->free_events.push_back(k->events.back());
            k->events.pop_back();
        }
    }
    if (host.extra) {
//  The loop ended inside the last batch: back to the state of its beginning, then only the loop's passes.
        for (size_t s = 0; s < k->item.setters.size(); s++) {
            void *target = ctx->buffers[k->input_keys[k->item.setters[s].input]].pointer;
            GFHIP_TRY(ctx, hipMemcpyAsync(target, k->undo[s], k->num_rays*esize, hipMemcpyDeviceToDevice, ctx->stream),
                      "hipMemcpyAsync(undo)");
        }
        if (launch_batch(k, host.batch_passes, nullptr)) return 1;
        GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, 8*sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
        GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    if (iterations_out) *iterations_out = static_cast<size_t> (host.iterations);
    if (last_max) *last_max = host.max_residual;
    return 0;
}

extern "C" int gfhip_converge(gfhip_kernel *k, double tolerance, size_t max_iterations,
                              size_t *iterations, double *last_max) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (k->output_keys.empty()) return ctx->fail("converge item has no output to reduce");
    if (settle(ctx)) return 1;
    if (!k->built) return ctx->fail("kernel has not been compiled (gfhip_compile)");
    if (!k->bound) return ctx->fail("kernel arguments are not bound (gfhip_create_kernel_call)");
    size_t used = 0;
    double residual = 0.0;
    int status;
    if (k->item.is_complex()) {
        status = k->item.base_is_f64() ? converge_loop_complex<double> (k, tolerance, max_iterations, &used, &residual)
                                       : converge_loop_complex<float> (k, tolerance, max_iterations, &used, &residual);
    } else if (k->batch_function && k->num_rays > 0) {
        status = converge_batched(k, tolerance, max_iterations, &used, &residual);
    } else if (k->max_function && k->num_rays > 0) {
        status = converge_on_device(k, tolerance, max_iterations, &used, &residual);
    } else if (k->item.dtype == GFIR_F64) {
        status = converge_loop<double> (k, tolerance, max_iterations, &used, &residual);
    } else {
        status = converge_loop<float> (k, tolerance, max_iterations, &used, &residual);
    }
    if (status) return status;
    if (iterations) *iterations = used;
    if (last_max) *last_max = residual;
    if (used > max_iterations) {
//  Same report as workflow.hpp:197-204.
        std::fprintf(stderr, "Workitem failed to converge with in given iterations.\nMinimum residual reached: %g\n", residual);
    }
    return 0;
}

//  Per-ray converge loop inside one launch (`<name>_converge`, see codegen.hpp).
extern "C" int gfhip_converge_per_ray(gfhip_kernel *k, double tolerance, size_t max_iterations,
                                      size_t *iterations, double *last_max) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    if (!k->built) return ctx->fail("kernel has not been compiled (gfhip_compile)");
    if (!k->converge_function) return ctx->fail("item has no setter/output to converge on");
    if (!k->bound) return ctx->fail("kernel arguments are not bound (gfhip_create_kernel_call)");
    if (k->num_rays == 0) {
        if (iterations) *iterations = 0;
        if (last_max) *last_max = -std::numeric_limits<double>::infinity();
        return 0;
    }
    std::vector<void *> pointers;
    for (auto key : k->input_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (auto key : k->output_keys) pointers.push_back(ctx->buffers[key].pointer);
    for (void *p : k->pack_device) pointers.push_back(p);
    pointers.push_back(ctx->device_flags);
    unsigned long long n = k->num_rays;
    double tolerance64 = tolerance;
    float tolerance32 = static_cast<float> (tolerance);
    unsigned int maximum = static_cast<unsigned int> (max_iterations > 0xFFFFFFFEull ? 0xFFFFFFFEull : max_iterations);
//  The iteration counter shares the 8-byte reduction scalar (low word).
    unsigned int *counter = reinterpret_cast<unsigned int *> (ctx->device_scalar);
    GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
    std::vector<void *> params;
    for (auto &p : pointers) params.push_back(&p);
    params.push_back(&n);
    params.push_back(k->item.dtype == GFIR_F64 ? static_cast<void *> (&tolerance64) : static_cast<void *> (&tolerance32));
    params.push_back(&maximum);
    params.push_back(&counter);
    GFHIP_TRY(ctx, hipModuleLaunchKernel(k->converge_function, k->grid, 1, 1, k->low.block_size, 1, 1,
                                         static_cast<unsigned int> (k->low.lds_bytes), ctx->stream,
                                         params.data(), nullptr), "hipModuleLaunchKernel(converge)");
    unsigned int used = 0;
    GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_scalar, ctx->device_scalar, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    used = static_cast<unsigned int> (*ctx->host_scalar & 0xFFFFFFFFull);
    if (iterations) *iterations = used;
    if (last_max) {
        const buffer &b = ctx->buffers[k->output_keys.back()];
        GFHIP_TRY(ctx, hipMemsetAsync(ctx->device_scalar, 0, sizeof(unsigned long long), ctx->stream), "hipMemsetAsync");
        gfhip::launch_max_reduce(b.pointer, k->num_rays, k->item.dtype == GFIR_F64, ctx->device_scalar,
                                 ctx->num_cus, ctx->stream);
        GFHIP_TRY(ctx, hipMemcpyAsync(ctx->host_scalar, ctx->device_scalar, sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync");
        GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        *last_max = decode_ordered(*ctx->host_scalar, k->item.dtype == GFIR_F64);
    }
    return 0;
}

extern "C" int gfhip_wait(gfhip_context *ctx) {
    if (!ctx) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
//  Host mirrors handed out by gfhip_get_host_buffer hold the device contents as of this drain:
//  all copies are queued behind the kernels, then ONE synchronisation.
    for (auto &kv : ctx->buffers) {
        buffer &b = kv.second;
        if (b.mirror && b.count) {
            GFHIP_TRY(ctx, hipMemcpyAsync(b.mirror, b.pointer, b.count*element_bytes(b.dtype),
                                          hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(mirror)");
        }
    }
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return 0;
}

extern "C" int gfhip_get_flags(gfhip_context *ctx, unsigned int *flags) {
    if (!ctx || !flags) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    GFHIP_TRY(ctx, hipMemcpy(flags, ctx->device_flags, sizeof(unsigned int), hipMemcpyDeviceToHost), "hipMemcpy(flags)");
    return 0;
}

static buffer *find_buffer(gfhip_context *ctx, const uint64_t key) {
    auto found = ctx->buffers.find(key);
    if (found == ctx->buffers.end()) {
        ctx->error = "unknown buffer key";
        return nullptr;
    }
    return &found->second;
}

extern "C" int gfhip_copy_to_device(gfhip_context *ctx, uint64_t key, const void *host) {
    if (!ctx) return 1;
    buffer *b = find_buffer(ctx, key);
    if (!b) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    GFHIP_TRY(ctx, hipMemcpyAsync(b->pointer, host, b->count*element_bytes(b->dtype),
                                  hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D)");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return 0;
}

extern "C" int gfhip_copy_to_host(gfhip_context *ctx, uint64_t key, void *host) {
    if (!ctx) return 1;
    buffer *b = find_buffer(ctx, key);
    if (!b) return 1;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    GFHIP_TRY(ctx, hipMemcpyAsync(host, b->pointer, b->count*element_bytes(b->dtype),
                                  hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(D2H)");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return 0;
}

extern "C" int gfhip_read_element(gfhip_context *ctx, uint64_t key, size_t index, void *element) {
    if (!ctx || !element) return 1;
    buffer *b = find_buffer(ctx, key);
    if (!b) return 1;
    if (index >= b->count) return ctx->fail("index out of range");
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    const size_t esize = element_bytes(b->dtype);
    GFHIP_TRY(ctx, hipMemcpy(element, static_cast<char *> (b->pointer) + index*esize, esize, hipMemcpyDeviceToHost), "hipMemcpy");
    return 0;
}

extern "C" int gfhip_check_value(gfhip_context *ctx, uint64_t key, size_t index, double *value) {
    if (!ctx || !value) return 1;
    double parts[2] = {0.0, 0.0};
    float narrow[2] = {0.0f, 0.0f};
    buffer *b = find_buffer(ctx, key);
    if (!b) return 1;
    const bool wide = b->dtype == GFIR_F64 || b->dtype == GFIR_C64;
    if (gfhip_read_element(ctx, key, index, wide ? static_cast<void *> (parts) : static_cast<void *> (narrow))) return 1;
    *value = wide ? parts[0] : static_cast<double> (narrow[0]);        // the real part of a complex element
    return 0;
}

extern "C" int gfhip_set_random_state(gfhip_kernel *k, uint64_t key, const void *states, size_t bytes) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    if (!k->item.has_random()) return 0;
//  The kernel indexes 1024 states of 2500 bytes (random.hpp:44-52 without the CUDA padding).
    const size_t needed = 1024*2500;
    if (!states || bytes < needed) return ctx->fail("a random state of 1024 MT19937 states (2500 bytes each) is required");
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    if (settle(ctx)) return 1;
    auto found = ctx->random_states.find(key);
    if (found == ctx->random_states.end()) {
        void *device = nullptr;
        GFHIP_TRY(ctx, hipMalloc(&device, bytes), "hipMalloc(random state)");
        const hipError_t status = hipMemcpy(device, states, bytes, hipMemcpyHostToDevice);
        if (status != hipSuccess) {
            (void)hipFree(device);
            return ctx->check(status, "hipMemcpy(random state)");
        }
        found = ctx->random_states.insert({key, {device, bytes}}).first;
    }
    k->random_states = found->second.first;
    return 0;
}

extern "C" void *gfhip_get_buffer(gfhip_context *ctx, uint64_t key, size_t *count) {
    if (!ctx) return nullptr;
    buffer *b = find_buffer(ctx, key);
    if (!b) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || settle(ctx)) return nullptr;      // the caller is about to look at the device buffer
    if (count) *count = b->count;
    return b->pointer;
}

extern "C" int gfhip_allocate_buffer(gfhip_context *ctx, uint64_t key, size_t count, uint32_t dtype) {
    if (!ctx) return 1;
    if (dtype > GFIR_C64) return ctx->fail("bad dtype");
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    return ensure_buffer(ctx, key, count, dtype, nullptr);
}

extern "C" int gfhip_get_buffer_info(gfhip_context *ctx, uint64_t key, size_t *count, uint32_t *dtype) {
    if (!ctx) return 1;
    buffer *b = find_buffer(ctx, key);
    if (!b) return 1;
    if (count) *count = b->count;
    if (dtype) *dtype = b->dtype;
    return 0;
}

extern "C" void *gfhip_get_host_buffer(gfhip_context *ctx, uint64_t key, size_t *count) {
    if (!ctx) return nullptr;
    buffer *b = find_buffer(ctx, key);
    if (!b) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || settle(ctx)) return nullptr;
    const size_t bytes = b->count*element_bytes(b->dtype);
    if (!b->mirror) {
        if (ctx->check(hipHostMalloc(&b->mirror, bytes ? bytes : 8, hipHostMallocDefault), "hipHostMalloc(mirror)")) {
            b->mirror = nullptr;
            return nullptr;
        }
        if (ctx->check(hipMemcpyAsync(b->mirror, b->pointer, bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync(mirror)") ||
            ctx->check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) {
            return nullptr;
        }
    }
    if (count) *count = b->count;
    return b->mirror;
}

extern "C" int gfhip_set_buffer(gfhip_context *ctx, uint64_t key, void *device_pointer, size_t count, uint32_t dtype) {
    if (!ctx) return 1;
    if (settle(ctx)) return 1;
    if (!device_pointer && count) return ctx->fail("null device pointer");
    if (dtype > GFIR_C64) return ctx->fail("bad dtype");
    auto found = ctx->buffers.find(key);
    if (found != ctx->buffers.end()) {
        if (found->second.owned && found->second.pointer) (void)hipFree(found->second.pointer);
        if (found->second.mirror) (void)hipHostFree(found->second.mirror);
    }
    buffer b;
    b.pointer = device_pointer;
    b.count = count;
    b.dtype = dtype;
    b.owned = false;
    ctx->buffers[key] = b;
    return 0;
}

extern "C" int gfhip_kernel_get_info(const gfhip_kernel *k, struct gfhip_kernel_info *info) {
    if (!k || !info) return 1;
    std::memset(info, 0, sizeof(*info));
    info->dtype = k->item.dtype;
    info->num_inputs = static_cast<uint32_t> (k->item.symbols.size());
    info->num_outputs = static_cast<uint32_t> (k->item.outputs.size());
    info->num_setters = static_cast<uint32_t> (k->item.setters.size());
    info->num_tables = static_cast<uint32_t> (k->item.tables.size());
    info->num_instructions = static_cast<uint32_t> (k->item.code.size());
    info->vgprs = static_cast<uint32_t> (k->vgprs);
    size_t lds = k->low.lds_bytes;
    for (auto &piece : k->pieces) lds = std::max(lds, piece.low.lds_bytes);
    info->lds_bytes = static_cast<uint32_t> (k->lds_static + lds);
    info->segments = static_cast<uint32_t> (k->pieces.size());
    info->converge_batch = k->batch_function ? k->low.batch : 0;          // segments the item runs as (0: one kernel)
    info->scratch_bytes = static_cast<uint32_t> (k->scratch);
    info->block_size = k->low.block_size;
    info->grid_size = k->grid;
    info->from_cache = k->from_cache ? 1 : 0;
    info->source_hash = k->low.hash;
    std::strncpy(info->name, k->low.kernel_name.c_str(), sizeof(info->name) - 1);
    return 0;
}

extern "C" int gfhip_enable_timing(gfhip_context *ctx, int enable) {
    if (!ctx) return 1;
    ctx->timing = enable > 0 ? static_cast<unsigned int> (enable) : 0;
    return 0;
}

extern "C" int gfhip_kernel_timing(gfhip_kernel *k, double *average_ms, uint64_t *launches) {
    if (!k) return 1;
    gfhip_context *ctx = k->ctx;
    GFHIP_TRY(ctx, hipSetDevice(ctx->device), "hipSetDevice");
    GFHIP_TRY(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    double total = 0.0;
    k->samples.clear();
    for (auto &e : k->events) {
        float ms = 0.0f;
        GFHIP_TRY(ctx, hipEventElapsedTime(&ms, e.first, e.second), "hipEventElapsedTime");
        total += ms;
        k->samples.push_back(ms);
        k->free_events.push_back(e);
    }
    if (launches) *launches = k->events.size();
    if (average_ms) *average_ms = k->events.empty() ? 0.0 : total/static_cast<double> (k->events.size());
    k->events.clear();
    return 0;
}

extern "C" int gfhip_kernel_timing_samples(gfhip_kernel *k, double *ms, size_t capacity, size_t *count) {
    if (!k) return 1;
    if (!k->events.empty() && gfhip_kernel_timing(k, nullptr, nullptr)) return 1;
    const size_t n = k->samples.size() < capacity ? k->samples.size() : capacity;
    if (ms && n) std::memcpy(ms, k->samples.data(), n*sizeof(double));
    if (count) *count = k->samples.size();
    return 0;
}

/* ---------------------------------------------------------------------------
 * gf_hip.h — C ABI of libgf_hip.so, the MI355X (gfx950) backend for
 * graph_framework work items.
 *
 * This is the drop-in boundary.  Every entry point replaces one member of the
 * reference's duck-typed backend context (the thing jit::context<T,SAFE_MATH>
 * forwards to: graph_framework/jit.hpp:63-74, :87-338; models
 * gpu::cpu_context cpu_context.hpp:82-611 and gpu::cuda_context
 * cuda_context.hpp:73-1005).  graph_framework_amd/hip_context.hpp is the thin
 * C++ class with the reference's member names that calls these functions; it
 * is what a maintainer adds next to cuda_context.hpp.
 *
 * The reference hands its backend a work item as generated C++ text plus the
 * node lists; this backend takes the node DAG itself, serialized as GFIR
 * (include/gfir.h), and lowers it to a CDNA4 kernel: one wavefront lane per
 * ray/particle, SoA state, spline coefficient tables re-laid-out AoS per cell
 * and staged through LDS where they fit.
 *
 * Conventions: plain pointers and sizes only.  Functions returning int return
 * 0 on success, non-zero on failure with a message in gfhip_last_error().
 * A context is bound to one device and one stream and is not thread safe
 * (one context per host thread/rank, as in the reference).  Buffers are keyed
 * by an opaque 64-bit value chosen by the caller (the reference keys them by
 * leaf_node*, cpu_context.hpp:87-89, cuda_context.hpp:78-80) and are shared by
 * all kernels of the context.
 * ------------------------------------------------------------------------- */
#ifndef GF_HIP_H
#define GF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfhip_context gfhip_context;
typedef struct gfhip_kernel gfhip_kernel;

/* Number of devices = number of host threads/ranks a driver should start.
 * Replaces  static size_t max_concurrency()   (cuda_context.hpp:121-125, jit.hpp:87). */
int gfhip_max_concurrency(void);

/* Replaces  static std::string device_type()  (cuda_context.hpp:130-132, jit.hpp:92). */
const char *gfhip_device_type(void);

/* Host side, no device: the contiguous split every reference driver applies to its ensemble, one
 * shard per device thread:  batch = total/shards, extra = total%shards, shard `index` holds
 * batch + (extra > index ? 1 : 0) elements  (graph_benchmark/xrays_bench.cpp:38-51,
 * graph_driver/xrays.cpp:423-432, graph_korc/xkorc.cpp:20-25).  Writes [*begin, *end) of the
 * shard; returns 1 if shards == 0 or index >= shards. */
int gfhip_shard_bounds(size_t total, size_t shards, size_t index, size_t *begin, size_t *end);

/* Bind device `index`.  `stream` is a hipStream_t to launch on, or NULL to
 * create a private one.  Replaces the context constructor  ctx(const size_t index)
 * (cuda_context.hpp:137-148, jit.hpp:101).  Returns NULL on failure
 * (gfhip_last_error(NULL) has the reason). */
gfhip_context *gfhip_create_context(int index, void *stream);

/* Frees every buffer, module and stream the context owns.
 * Replaces ~cuda_context (cuda_context.hpp:153-185). */
void gfhip_destroy_context(gfhip_context *ctx);

/* Message of the last failure on this context (ctx may be NULL for creation errors). */
const char *gfhip_last_error(const gfhip_context *ctx);

/* Register one work item.  `gfir`/`bytes` is the serialized item (include/gfir.h);
 * `num_rays` is the ensemble size it runs over.
 * Replaces jit::context::add_kernel (jit.hpp:118-194) together with the
 * backend's create_kernel_prefix/create_kernel_postfix
 * (cuda_context.hpp:713-946, cpu_context.hpp:428-584). */
gfhip_kernel *gfhip_add_kernel(gfhip_context *ctx, const void *gfir, size_t bytes, size_t num_rays);

/* Lower and build every kernel added so far (pre-built code objects are looked
 * up by source hash in the kernel cache directories, otherwise hipRTC).
 * Replaces  void compile(source, names, add_reduction)  (cuda_context.hpp:194-302, jit.hpp:238-244). */
int gfhip_compile(gfhip_context *ctx);

/* Bind the kernel's arguments to buffers.  input_keys[i] / output_keys[o] name
 * the buffers; on first sight of a key the buffer is allocated with num_rays
 * elements (zero-filled) and, for inputs with a non-NULL input_init[i], its first
 * input_counts[i] elements (num_rays if input_counts is NULL) are filled from
 * that host array.  Replaces  create_kernel_call(name, inputs, outputs, state,
 * num_rays, ...)  (cuda_context.hpp:316-531, cpu_context.hpp:233-298). */
int gfhip_create_kernel_call(gfhip_kernel *kernel,
                             const uint64_t *input_keys, const void *const *input_init,
                             const size_t *input_counts, const uint64_t *output_keys);

/* Asynchronous launch of the kernel on the context's stream (the closure
 * create_kernel_call returns in the reference).  `steps` > 1 repeats the item
 * inside one launch, keeping the state in registers between passes. */
int gfhip_run(gfhip_kernel *kernel, uint32_t steps);

/* Run the kernel, reduce max over its LAST output on the device, synchronise and
 * return the scalar.  Items of up to 1500 nodes reduce inside the launch
 * (`<name>_max`: wave shuffle + one atomic per workgroup); larger ones run the
 * separate reduction kernel over the output buffer.  Replaces
 * create_max_call(arg, run)  (cuda_context.hpp:540-576, cpu_context.hpp:306-322).
 * Called again and again with no other entry point in between — which is what the
 * reference's converge_item::run (workflow.hpp:179-205) does through that closure — it
 * runs ahead: from the second call of the streak on, one `<name>_batch` launch runs
 * the asked pass and the next ones, each with its own max, the following calls are
 * answered from those without a launch or a synchronisation, and the next entry point
 * of any kind first takes back the passes nobody asked for (state of the beginning of
 * the batch restored, the asked passes run again).  Same values, same state, fewer
 * launches (10 instead of 25 for the benchmark's Newton solve); GFHIP_RUN_AHEAD=0
 * keeps one launch per call. */
int gfhip_run_max(gfhip_kernel *kernel, double *max_value);

/* The same for items of a complex type (enum gfir_dtype GFIR_C32 / GFIR_C64): value[0] + i value[1] =
 * the element of largest modulus of the last output, the first of equals, as
 * cpu_context.hpp:314-318 selects it (std::max_element on std::abs). */
int gfhip_run_max_complex(gfhip_kernel *kernel, double *value);

/* Reduce a BUFFER, whatever wrote it: value[0] (+ i value[1] for a complex buffer) = the max of its
 * elements as cpu_context takes it (std::max_element, cpu_context.hpp:306-322: a NaN only if it is
 * element 0; complex: the element of largest modulus, the first of equals), after everything queued on
 * the context's stream; synchronises.  This is the general form of  create_max_call(arg, run)
 * (jit.hpp:274-277): the caller runs `run` and then reduces `arg`'s buffer, which need not be the last
 * output of the kernel that `run` launches. */
int gfhip_reduce_max(gfhip_context *ctx, uint64_t key, double *value);

/* The loop of workflow::converge_item::run (workflow.hpp:179-205) around
 * gfhip_run_max: repeat until |max| <= tol, or max stalls against the previous
 * or the previous-but-one value, or max_iterations.  The loop's test runs on the
 * device after each pass and passes are queued ahead of the host, so the host
 * synchronises once per batch of passes; the passes that run, the iteration
 * count and the results are identical to calling gfhip_run_max from that loop. */
int gfhip_converge(gfhip_kernel *kernel, double tolerance, size_t max_iterations,
                   size_t *iterations, double *last_max);

/* The same stall loop run PER RAY inside one launch: each lane iterates on its own residual
 * and a wavefront leaves the loop when the ballot of active lanes is empty.  This is the
 * reference loop applied to every ray as its own shard: identical to gfhip_converge when the
 * rays are identical (the benchmark), otherwise rays stop as soon as they have stalled
 * instead of iterating until the slowest ray of the shard has.  `iterations` receives the
 * maximum over rays, `last_max` the maximum final residual. */
int gfhip_converge_per_ray(gfhip_kernel *kernel, double tolerance, size_t max_iterations,
                           size_t *iterations, double *last_max);

/* Drain the stream.  Replaces  void wait()  (cuda_context.hpp:581-584). */
int gfhip_wait(gfhip_context *ctx);

/* Status bits raised by kernels since the context was created (after a drain);
 * informational: the lanes concerned redid their pass with the compiler's IEEE
 * division, results are the IEEE ones either way.
 * Bit 0: a lane failed a check of the shared-reciprocal division (fp64: a
 *        denominator outside [2^-500, 2^500], a non-finite result or gather index
 *        quotient; fp32: a zero, infinite or NaN denominator).
 * Bit 1: fp64 only: a lane stored a zero computed from a quotient (its sign is only
 *        the IEEE one through v_div_fixup). */
int gfhip_get_flags(gfhip_context *ctx, unsigned int *flags);

/* Whole-buffer copies, synchronous on return.  Replace copy_to_device /
 * copy_to_host (cuda_context.hpp:625-643; callers read host data right after,
 * dispersion.hpp:1472). */
int gfhip_copy_to_device(gfhip_context *ctx, uint64_t key, const void *host);
int gfhip_copy_to_host(gfhip_context *ctx, uint64_t key, void *host);

/* Read one element after draining the stream.  Replaces  T check_value(index, node)
 * (cuda_context.hpp:602-607).  gfhip_check_value returns it as a double (the real part of a
 * complex element); gfhip_read_element copies the element itself (4, 8 or 16 bytes). */
int gfhip_check_value(gfhip_context *ctx, uint64_t key, size_t index, double *value);
int gfhip_read_element(gfhip_context *ctx, uint64_t key, size_t index, void *element);

/* Bind the MT19937 states of the item's random_state node (random.hpp:24-130): `states` are
 * the node's 1024 mt_state structures (state->data(), 2500 bytes each), uploaded on first
 * sight of `key` and shared by every kernel bound to that key, as create_kernel_call does
 * (cuda_context.hpp:367-380).  No-op for items without a random node. */
int gfhip_set_random_state(gfhip_kernel *kernel, uint64_t key, const void *states, size_t bytes);

/* Device pointer and element count of a buffer (NULL if the key is unknown).
 * The reference's get_buffer (cuda_context.hpp:650-652) returns the managed
 * pointer; here it is device memory. */
void *gfhip_get_buffer(gfhip_context *ctx, uint64_t key, size_t *count);

/* Allocate (zero-filled) the buffer for `key` if it does not exist yet: the
 * buffer side of create_kernel_call for nodes the kernel itself never stores
 * (an output that is a variable or equals a setter's expression,
 * cpu_context.hpp:551-553, still gets its buffer at cuda_context.hpp:364-383). */
int gfhip_allocate_buffer(gfhip_context *ctx, uint64_t key, size_t count, uint32_t dtype);

/* Element count and dtype (enum gfir_dtype) of a buffer; non-zero if the key is unknown. */
int gfhip_get_buffer_info(gfhip_context *ctx, uint64_t key, size_t *count, uint32_t *dtype);

/* A stable, pinned HOST copy of the buffer for `key`, valid for the life of the
 * context and refreshed by every gfhip_wait() (all mirrors are copied behind
 * the queued kernels, then one synchronisation).  This is what the reference's
 *   T *get_buffer(node)   (jit.hpp:336, cuda_context.hpp:1002-1004; managed
 * memory there) means to its caller output.hpp:271: a pointer the NetCDF
 * writer thread reads after wait(). */
void *gfhip_get_host_buffer(gfhip_context *ctx, uint64_t key, size_t *count);

/* Adopt caller-owned device memory (e.g. a shard of a larger allocation) as the
 * buffer for `key`; the context will not free it. */
int gfhip_set_buffer(gfhip_context *ctx, uint64_t key, void *device_pointer, size_t count, uint32_t dtype);

/* Introspection used by hosts, tests and the benchmark. */
struct gfhip_kernel_info {
    uint32_t dtype;                 /* enum gfir_dtype */
    uint32_t num_inputs, num_outputs, num_setters, num_tables, num_instructions;
    uint32_t vgprs, agprs, sgprs, lds_bytes, scratch_bytes;   /* from the code object, after compile */
    uint32_t block_size, grid_size; /* launch geometry for num_rays */
    uint32_t from_cache;            /* 1 if the code object came from the kernel cache */
    uint32_t segments;              /* kernels the item runs as when it was cut into segments, else 0 */
    uint32_t converge_batch;        /* passes per launch of gfhip_converge's loop (`<name>_batch`), 0 or 1: one launch per pass */
    uint32_t reserved;
    uint64_t source_hash;
    char     name[64];
};
int gfhip_kernel_get_info(const gfhip_kernel *kernel, struct gfhip_kernel_info *info);

/* Lowering without a device: returns the generated HIP source of one item in a
 * malloc'ed, NUL-terminated string (free with gfhip_free_string) and its hash.
 * Used by __graft_entry__.build() to pre-build code objects with hipcc. */
char *gfhip_generate_source(const void *gfir, size_t bytes, uint64_t *source_hash);

/* The same for an item that runs as several kernels: items above GFHIP_SEGMENT_NODES records (default
 * 6000; e.g. a ray step on the 86-mode VMEC equilibrium, equilibrium.hpp:1868-2330: 54 k records) are
 * cut into consecutive segments, each lowered and cached as a translation unit of its own.  *source =
 * the text of piece `index` (free with gfhip_free_string), or NULL past the last piece; an item that
 * runs as one kernel has exactly one piece.  Returns non-zero on a malformed item. */
int gfhip_generate_piece_source(const void *gfir, size_t bytes, uint32_t index, char **source, uint64_t *source_hash);

/* Piece `index` of a segmented item as data, for checking the split without a device: *piece = a
 * malloc'ed block (free with gfhip_free_string) of int32 words
 *   { symbols S, outputs O, hand-over slots, pieces,
 *     S x state input it reads (-1: none), S x hand-over slot it reads (-1: none),
 *     O x hand-over slot it writes (-1: none), O x output of the item it is (-1: none) }
 * followed by the piece as a GFIR item (include/gfir.h).  *piece stays NULL past the last piece and
 * for items that run as one kernel. */
int gfhip_export_piece(const void *gfir, size_t bytes, uint32_t index, void **piece, size_t *piece_bytes);
void gfhip_free_string(char *text);

/* Host side, no device: the initial conditions of the xrays command line for one shard, sample
 * for sample (graph_driver/xrays.cpp:397-453: std::mt19937_64(seed = shard index), libstdc++'s
 * std::normal_distribution per variable, drawn in the order omega, kx, ky, kz, z, (x, y)).
 * means/sigmas: 7 doubles each in that order with (radius, phi) last, sigma <= 0 = the mean, no
 * draw; columns: 8 arrays of n doubles, t, w, x, y, z, kx, ky, kz. */
void gfhip_cli_distribution(uint64_t seed, size_t n, const double *means, const double *sigmas, double *const *columns);

/* Average duration in milliseconds of the launches of `kernel` recorded since
 * the last call: HIP events on the context's stream around every `enable`-th
 * launch of each kernel (1 = every launch, 0 = off; a pair of event records
 * costs the stream 2-8 us, so a benchmark samples).  Used by bench.py for the
 * roofline line. */
int gfhip_enable_timing(gfhip_context *ctx, int enable);
int gfhip_kernel_timing(gfhip_kernel *kernel, double *average_ms, uint64_t *launches);
/* The individual durations (ms, launch order) behind the last gfhip_kernel_timing
 * (which this call performs first if launches are pending): up to `capacity`
 * values into `ms`, the number available into `count`. */
int gfhip_kernel_timing_samples(gfhip_kernel *kernel, double *ms, size_t capacity, size_t *count);

#ifdef __cplusplus
}
#endif

#endif /* GF_HIP_H */

//------------------------------------------------------------------------------
///  @file hip_context.hpp
///  @brief gpu::hip_context — the MI355X backend context for graph_framework.
///
///  Drop this header next to graph_framework/cuda_context.hpp and select it in
///  jit.hpp with -DUSE_HIP (INTEGRATION.md shows the three-line patch).  It has
///  the duck-typed interface jit::context<T, SAFE_MATH> forwards to
///  (graph_framework/jit.hpp:63-74, :87-338; same members as gpu::cpu_context,
///  cpu_context.hpp:82-611, and gpu::cuda_context, cuda_context.hpp:73-1005).
///
///  Unlike the CPU/CUDA/Metal contexts it does not compile the C++ text the
///  nodes write: create_kernel_prefix/create_kernel_postfix receive the work
///  item's node lists, the DAG is serialized to GFIR (gfir_serialize.hpp) and
///  handed to libgf_hip.so (include/gf_hip.h), which lowers it to a gfx950
///  kernel.  The text stream is still fed (registers must exist for the nodes'
///  compile() methods to run) but is otherwise ignored.
///
///  All eight flavours of the reference are served: float, double and their
///  complex forms, with and without the SAFE_MATH guards (cpu_context.hpp:530-547,
///  arithmetic.hpp:2534-2557), kernels with a random state (random.hpp) and erfi
///  nodes (math.hpp:1440; csrc/prelude.hpp gf_erfi).
//------------------------------------------------------------------------------
#ifndef hip_context_h
#define hip_context_h

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "random.hpp"
//  The node classes the serializer walks (jit.hpp itself only needs node.hpp).
#include "arithmetic.hpp"
#include "math.hpp"
#include "trigonometry.hpp"
#include "piecewise.hpp"

#include "../include/gf_hip.h"
#include "gfir_serialize.hpp"

namespace gpu {
//------------------------------------------------------------------------------
///  @brief Class representing a HIP gpu context.
///
///  @tparam T         Base type of the calculation.
///  @tparam SAFE_MATH Use @ref general_concepts_safe_math operations.
//------------------------------------------------------------------------------
    template<jit::float_scalar T, bool SAFE_MATH=false>
    class hip_context {
    private:
///  Handle of the C ABI context (one device, one stream).
        gfhip_context *context;

///  A work item between create_kernel_prefix and compile.
        struct pending_item {
            std::string name;
            graph::input_nodes<T, SAFE_MATH> inputs;
            graph::output_nodes<T, SAFE_MATH> outputs;
            size_t size;
            std::vector<uint8_t> gfir;
        };
        std::vector<pending_item> pending;
///  What the kernel really takes, after the de-duplication the reference's prefix/postfix do
///  (each node is bound once, cuda_context.hpp:330-383; an output that is a variable, that equals
///  a setter's expression or that was listed before is not stored again, cpu_context.hpp:551-553).
        struct bound_kernel {
            gfhip_kernel *kernel;
            graph::input_nodes<T, SAFE_MATH> inputs;        ///< distinct input variables, first-seen order
            graph::output_nodes<T, SAFE_MATH> outputs;      ///< outputs the kernel stores
///  Expression node -> the variable whose buffer holds its value after the kernel: a setter's
///  expression is stored into the setter's variable and, listed as an output as well, not again.
            std::map<graph::leaf_node<T, SAFE_MATH> *, graph::leaf_node<T, SAFE_MATH> *> held_by;
        };
        std::map<std::string, bound_kernel> kernels;

///  The closure create_kernel_call returns.  A type of its own, so that create_max_call can tell
///  from its `run` argument (std::function::target) which kernel it is asked to run.
        struct kernel_call {
            hip_context *self;
            gfhip_kernel *kernel;
            const bound_kernel *bound;
            void operator()() const {
                self->check(gfhip_run(kernel, 1), "gfhip_run");
            }
        };

        static uint64_t key(graph::leaf_node<T, SAFE_MATH> *node) {
            return static_cast<uint64_t> (reinterpret_cast<uintptr_t> (node));
        }

        static T from_parts(const double *value) {
            if constexpr (jit::complex_scalar<T>) {
                return T(static_cast<typename T::value_type> (value[0]), static_cast<typename T::value_type> (value[1]));
            } else {
                return static_cast<T> (value[0]);
            }
        }

        void check(const int status, const char *what) const {
            if (status) {
                std::cerr << "hip_context: " << what << ": " << gfhip_last_error(context) << std::endl;
                exit(-1);
            }
        }

    public:
///  Size of random state needed.
        constexpr static size_t random_state_size = 1024;

///  Remaining constant memory in bytes (tables live in packed global/LDS arrays).
        int remaining_const_memory;

//------------------------------------------------------------------------------
///  @brief Get the maximum number of concurrent instances.
//------------------------------------------------------------------------------
        static size_t max_concurrency() {
            return static_cast<size_t> (gfhip_max_concurrency());
        }

//------------------------------------------------------------------------------
///  @brief Device discription.
//------------------------------------------------------------------------------
        static std::string device_type() {
            return gfhip_device_type();
        }

//------------------------------------------------------------------------------
///  @brief Construct a HIP context.
///
///  @param[in] index Concurrent index.
//------------------------------------------------------------------------------
        hip_context(const size_t index) : remaining_const_memory(0) {
            context = gfhip_create_context(static_cast<int> (index), nullptr);
            if (!context) {
                std::cerr << "hip_context: " << gfhip_last_error(nullptr) << std::endl;
                exit(-1);
            }
        }

        ~hip_context() {
            gfhip_destroy_context(context);
        }

//  The context owns device state: it moves (graph_set_device_number assigns a fresh
//  workflow::manager over the old one, graph_c_binding.cpp:1939-1980) but does not copy.
        hip_context(const hip_context &) = delete;
        hip_context &operator=(const hip_context &) = delete;
        hip_context(hip_context &&other) noexcept :
        context(other.context), pending(std::move(other.pending)), kernels(std::move(other.kernels)),
        remaining_const_memory(other.remaining_const_memory) {
            other.context = nullptr;
        }
        hip_context &operator=(hip_context &&other) noexcept {
            if (this != &other) {
                gfhip_destroy_context(context);
                context = other.context;
                other.context = nullptr;
                pending = std::move(other.pending);
                kernels = std::move(other.kernels);
                remaining_const_memory = other.remaining_const_memory;
            }
            return *this;
        }

//------------------------------------------------------------------------------
///  @brief Create the source header (nothing to declare: the text is not compiled).
//------------------------------------------------------------------------------
        void create_header(std::ostringstream &source_buffer) {
            source_buffer << "// hip_context lowers the node graph directly; this text is informational."
                          << std::endl;
        }

//------------------------------------------------------------------------------
///  @brief Begin a kernel: record its name, inputs, outputs and size.
///
///  Every input needs a register name because the nodes' compile() methods look
///  it up (cpu_context.hpp:489-500 does the same).
//------------------------------------------------------------------------------
        void create_kernel_prefix(std::ostringstream &source_buffer,
                                  const std::string name,
                                  graph::input_nodes<T, SAFE_MATH> &inputs,
                                  graph::output_nodes<T, SAFE_MATH> &outputs,
                                  graph::shared_random_state<T, SAFE_MATH> state,
                                  const size_t size,
                                  const std::vector<bool> &is_constant,
                                  jit::register_map &registers,
                                  const jit::register_usage &usage,
                                  jit::texture1d_list &textures1d,
                                  jit::texture2d_list &textures2d) {
            (void)is_constant; (void)usage; (void)textures1d; (void)textures2d;
            source_buffer << "// kernel " << name << std::endl;
            for (auto &input : inputs) {
                registers[input.get()] = jit::to_string('v', input.get());
            }
            if (state.get()) {
                registers[state.get()] = jit::to_string('s', state.get());      // random_node::compile reads it
            }
            pending_item item;
            item.name = name;
            for (auto &input : inputs) {
                if (std::find(item.inputs.begin(), item.inputs.end(), input) == item.inputs.end()) {
                    item.inputs.push_back(input);
                }
            }
            item.outputs = outputs;
            item.size = size;
            pending.push_back(item);
        }

//------------------------------------------------------------------------------
///  @brief End a kernel: the setters complete the work item; serialize it.
//------------------------------------------------------------------------------
        void create_kernel_postfix(std::ostringstream &source_buffer,
                                   graph::output_nodes<T, SAFE_MATH> &outputs,
                                   graph::map_nodes<T, SAFE_MATH> &setters,
                                   graph::shared_random_state<T, SAFE_MATH> state,
                                   jit::register_map &registers,
                                   jit::register_map &indices,
                                   const jit::register_usage &usage) {
            (void)outputs; (void)state; (void)registers; (void)indices; (void)usage;
            source_buffer << "// end kernel" << std::endl;
            {
                pending_item &item = pending.back();
//  The stores the reference's postfix would emit (cpu_context.hpp:522-580): setters whose
//  expression is not the variable itself, then outputs that are neither variables nor already
//  stored by a setter or an earlier output.
                graph::map_nodes<T, SAFE_MATH> stores;
                std::vector<graph::leaf_node<T, SAFE_MATH> *> stored;
                for (auto &[out, in] : setters) {
                    if (!out->is_match(in)) {
                        stores.push_back({out, in});
                        stored.push_back(out.get());
                    }
                }
                graph::output_nodes<T, SAFE_MATH> kept;
                for (auto &out : item.outputs) {
                    if (!graph::variable_cast(out).get() &&
                        std::find(stored.begin(), stored.end(), out.get()) == stored.end()) {
                        kept.push_back(out);
                        stored.push_back(out.get());
                    }
                }
                gfir::serializer<T, SAFE_MATH> serialize;
                const std::string text = source_buffer.str();
                serialize.kernel_text = &text;
                item.gfir = serialize(item.name, item.inputs, kept, stores);
                bound_kernel bound{nullptr, item.inputs, kept, {}};
                for (auto &[out, in] : stores) {
                    bound.held_by[out.get()] = in.get();
                }
                kernels[item.name] = bound;
            }
        }

//------------------------------------------------------------------------------
///  @brief Create a reduction (the device max reduction is part of libgf_hip).
//------------------------------------------------------------------------------
        void create_reduction(std::ostringstream &source_buffer, const size_t size) {
            (void)source_buffer; (void)size;
        }

//------------------------------------------------------------------------------
///  @brief Compile the kernels.
///
///  @param[in] kernel_source Source text (ignored).
///  @param[in] names         Names of the kernel functions.
///  @param[in] add_reduction Include the reduction kernel (always available).
//------------------------------------------------------------------------------
        void compile(const std::string kernel_source,
                     std::vector<std::string> names,
                     const bool add_reduction=false) {
            (void)kernel_source; (void)names; (void)add_reduction;
            for (auto &item : pending) {
                gfhip_kernel *kernel = gfhip_add_kernel(context, item.gfir.data(), item.gfir.size(), item.size);
                if (!kernel) {
                    check(1, "gfhip_add_kernel");
                }
                kernels[item.name].kernel = kernel;
            }
            pending.clear();
            check(gfhip_compile(context), "gfhip_compile");
        }

//------------------------------------------------------------------------------
///  @brief Create a kernel calling function.
///
///  Buffers are keyed by node; a node seen for the first time is allocated and,
///  for inputs, filled with node->evaluate() (cuda_context.hpp:316-383).
//------------------------------------------------------------------------------
        std::function<void(void)> create_kernel_call(const std::string kernel_name,
                                                     graph::input_nodes<T, SAFE_MATH> inputs,
                                                     graph::output_nodes<T, SAFE_MATH> outputs,
                                                     graph::shared_random_state<T, SAFE_MATH> state,
                                                     const size_t num_rays,
                                                     const jit::texture1d_list &tex1d_list,
                                                     const jit::texture2d_list &tex2d_list) {
            (void)inputs; (void)tex1d_list; (void)tex2d_list;
            auto found = kernels.find(kernel_name);
            if (found == kernels.end() || !found->second.kernel) {
                std::cerr << "hip_context: kernel " << kernel_name << " was not added and compiled." << std::endl;
                exit(-1);
            }
            bound_kernel &bound = found->second;
            gfhip_kernel *kernel = bound.kernel;

            std::vector<uint64_t> input_keys, output_keys;
            std::vector<backend::buffer<T>> initial;
            std::vector<const void *> initial_pointers;
            std::vector<size_t> initial_counts;
            for (auto &input : bound.inputs) {
                input_keys.push_back(key(input.get()));
                initial.push_back(input->evaluate());
            }
            for (auto &buffer : initial) {
                initial_pointers.push_back(buffer.data());
                initial_counts.push_back(buffer.size());
            }
            for (auto &output : bound.outputs) {
                output_keys.push_back(key(output.get()));
            }
            check(gfhip_create_kernel_call(kernel, input_keys.data(), initial_pointers.data(), initial_counts.data(),
                                           output_keys.data()), "gfhip_create_kernel_call");
//  Every listed node gets its buffer on first sight, stored by this kernel or not.
            const uint32_t dtype = jit::complex_scalar<T> ? (jit::float_base<T> ? GFIR_C32 : GFIR_C64)
                                                          : (jit::float_base<T> ? GFIR_F32 : GFIR_F64);
            if (state.get()) {
                check(gfhip_set_random_state(kernel, key(state.get()), state->data(), state->get_size_bytes()),
                      "gfhip_set_random_state");
            }
            for (auto &output : outputs) {
                check(gfhip_allocate_buffer(context, key(output.get()), num_rays, dtype), "gfhip_allocate_buffer");
            }
            return kernel_call{this, kernel, &bound};
        }

//------------------------------------------------------------------------------
///  @brief Create a max compute kernel calling function.
//------------------------------------------------------------------------------
        std::function<T(void)> create_max_call(graph::shared_leaf<T, SAFE_MATH> &argument,
                                               std::function<void(void)> run) {
//  The contract (jit.hpp:274-277; cuda_context.hpp:540-576, cpu_context.hpp:306-322): run `run`,
//  then reduce the buffer that holds `argument`.  `run` is normally the closure the preceding
//  create_kernel_call returned (workflow.hpp:172): when it is, and `argument` is the LAST output
//  that kernel stores, the max is folded into the kernel's own launch (gfhip_run_max — which, called in a row as
//  workflow.hpp:179-205 calls this closure, runs ahead: a batch of passes per launch, each with its own max, the
//  passes nobody asked for taken back by the next entry point; include/gf_hip.h).  In every
//  other case — a foreign `run`, an argument that is a variable, that a setter stores or that is
//  not the last output — `run` runs as given and the buffer holding `argument` is reduced.
            const kernel_call *call = run.template target<kernel_call> ();
            graph::leaf_node<T, SAFE_MATH> *holder = argument.get();
            if (call && call->self == this) {
                auto held = call->bound->held_by.find(holder);
                if (held != call->bound->held_by.end()) {
                    holder = held->second;
                }
                if (!call->bound->outputs.empty() && call->bound->outputs.back().get() == argument.get()) {
                    gfhip_kernel *kernel = call->kernel;
                    return [this, kernel] () mutable {
                        double value[2];
                        check(gfhip_run_max_complex(kernel, value), "gfhip_run_max");
                        return from_parts(value);
                    };
                }
            } else {
                for (auto &[name, bound] : kernels) {
                    auto held = bound.held_by.find(holder);
                    if (held != bound.held_by.end()) {
                        holder = held->second;
                        break;
                    }
                }
            }
            const uint64_t buffer_key = key(holder);
            return [this, run, buffer_key] () mutable {
                run();
                double value[2];
                check(gfhip_reduce_max(context, buffer_key, value), "gfhip_reduce_max");
                return from_parts(value);
            };
        }

//------------------------------------------------------------------------------
///  @brief Hold the current thread until the stream has completed; refresh host mirrors.
//------------------------------------------------------------------------------
        void wait() {
            check(gfhip_wait(context), "gfhip_wait");
        }

//------------------------------------------------------------------------------
///  @brief Print out the results.
//------------------------------------------------------------------------------
        void print_results(const size_t index,
                           const graph::output_nodes<T, SAFE_MATH> &nodes) {
            for (auto &out : nodes) {
                std::cout << check_value(index, out) << " ";
            }
            std::cout << std::endl;
        }

//------------------------------------------------------------------------------
///  @brief Check the value.
//------------------------------------------------------------------------------
        T check_value(const size_t index,
                      const graph::shared_leaf<T, SAFE_MATH> &node) {
            T value;
            check(gfhip_read_element(context, key(node.get()), index, &value), "gfhip_read_element");
            return value;
        }

//------------------------------------------------------------------------------
///  @brief Copy buffer contents to the device.
//------------------------------------------------------------------------------
        void copy_to_device(graph::shared_leaf<T, SAFE_MATH> node,
                            T *source) {
            check(gfhip_copy_to_device(context, key(node.get()), source), "gfhip_copy_to_device");
        }

//------------------------------------------------------------------------------
///  @brief Copy buffer contents to host (complete on return).
//------------------------------------------------------------------------------
        void copy_to_host(graph::shared_leaf<T, SAFE_MATH> node,
                          T *destination) {
            check(gfhip_copy_to_host(context, key(node.get()), destination), "gfhip_copy_to_host");
        }

//------------------------------------------------------------------------------
///  @brief Get a stable host-readable buffer for a node (output.hpp:271).
///
///  A pinned host mirror owned by the library: the pointer stays valid for the life of
///  the context and holds the device contents as of the last wait().
//------------------------------------------------------------------------------
        T *get_buffer(graph::shared_leaf<T, SAFE_MATH> &node) {
            void *mirror = gfhip_get_host_buffer(context, key(node.get()), nullptr);
            if (!mirror) {
                check(1, "gfhip_get_host_buffer");
            }
            return static_cast<T *> (mirror);
        }
    };
}

#endif /* hip_context_h */

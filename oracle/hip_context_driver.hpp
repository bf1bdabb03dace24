// ---------------------------------------------------------------------------
// hip_context_driver.hpp — TEST INFRASTRUCTURE.  The few lines of jit::context
// (graph_framework/jit.hpp) that touch a backend context, restated with citations so that
// gpu::hip_context can be driven exactly the way the reference drives its backends
// (jit.hpp itself needs the LLVM JIT headers and cannot be compiled in this image).
// Shared by hip_context_demo.cpp and hip_context_physics.cpp.
// ---------------------------------------------------------------------------
#ifndef hip_context_driver_hpp
#define hip_context_driver_hpp

#include "ref_builders.hpp"
#include "../graph_framework_amd/hip_context.hpp"

//  jit::context::add_kernel, jit.hpp:118-194: preamble pass, prefix, node text, postfix.
template<typename T, typename CONTEXT>
static void add_kernel(CONTEXT &gpu, std::ostringstream &source, jit::register_map &registers,
                       const std::string name,
                       graph::input_nodes<T> inputs, graph::output_nodes<T> outputs,
                       graph::map_nodes<T> setters, const size_t size) {
    std::vector<bool> is_constant(inputs.size(), true);
    jit::visiter_map visited;
    jit::register_usage usage;
    jit::texture1d_list textures1d;
    jit::texture2d_list textures2d;
    for (auto &[out, in] : setters) {
        auto found = std::distance(inputs.begin(), std::find(inputs.begin(), inputs.end(), in));
        if (static_cast<size_t> (found) < is_constant.size()) {
            is_constant[found] = false;
        }
        out->compile_preamble(source, registers, visited, usage, textures1d, textures2d,
                              gpu.remaining_const_memory);
    }
    for (auto &out : outputs) {
        out->compile_preamble(source, registers, visited, usage, textures1d, textures2d,
                              gpu.remaining_const_memory);
    }
    for (auto &in : inputs) {
        if (usage.find(in.get()) == usage.end()) {
            usage[in.get()] = 0;
        }
    }
    gpu.create_kernel_prefix(source, name, inputs, outputs, graph::shared_random_state<T> (), size,
                             is_constant, registers, usage, textures1d, textures2d);
    jit::register_map indices;
    for (auto &[out, in] : setters) {
        out->compile(source, registers, indices, usage);
    }
    for (auto &out : outputs) {
        out->compile(source, registers, indices, usage);
    }
    gpu.create_kernel_postfix(source, outputs, setters, graph::shared_random_state<T> (),
                              registers, indices, usage);
    std::vector<void *> removed;                                     // jit.hpp:184-193
    for (auto &[key, value] : registers) {
        if (value[0] == 'r') removed.push_back(key);
    }
    for (auto &key : removed) registers.erase(key);
}

template<typename T, typename ITEM>
static void to_lists(const ITEM &item, graph::input_nodes<T> &in, graph::map_nodes<T> &set) {
    for (auto &i : item.in_nodes) in.push_back(graph::variable_cast(i));
    for (auto &s : item.set_nodes) set.push_back({s.first, graph::variable_cast(s.second)});
}

#endif /* hip_context_driver_hpp */

// ---------------------------------------------------------------------------
// hip_context_nodes.cpp — small graphs built with the reference's node factories, run through
// gpu::hip_context (hence through gfir_serialize.hpp on whatever the reference's reducer made of
// them) and through the tape interpreter; the two must agree bit for bit (transcendentals: to
// the stated tolerance).  TEST INFRASTRUCTURE; builds to oracle/_ref/hip_context_nodes.
//
// The graphs are the kinds the reference's kernel-level tests exercise — gathers with arguments
// beyond both ends of the table (compile_index clamp, piecewise.hpp:26-65), tables folded by the
// reducer (piecewise +,-,*,/ piecewise or constant; 2-D combined with 1-D along a row or a
// column: graph_tests/piecewise_test.cpp:80-316, :319-830), every arithmetic and math node
// (graph_tests/jit_test.cpp, math_test.cpp, trigonometry_test.cpp), setters that read the old
// values (graph_tests/workflow_test.cpp:20-71) — but the expected values come from the tape
// interpreter evaluating the same DAG, for double and for float, over several rays at once.
//
// Usage: hip_context_nodes
// ---------------------------------------------------------------------------
#include "hip_context_driver.hpp"

#include <cstring>

namespace {

int failures = 0;
int cases = 0;

template<typename T>
bool same_bits(const T a, const T b) {
    return std::memcmp(&a, &b, sizeof(T)) == 0 || (a == b);       // +0 and -0 compare equal
}

//  One work item on both paths.  `values[i]` initialises input i (one entry per ray).
template<typename T>
void run_case(const char *name, const std::vector<leaf<T>> &inputs, const std::vector<std::vector<T>> &values,
              const std::vector<leaf<T>> &outputs, const std::vector<std::pair<leaf<T>, leaf<T>>> &setters,
              const T tolerance = 0, const int launches = 1) {
    const size_t n = values[0].size();
    for (size_t i = 0; i < inputs.size(); i++) graph::variable_cast(inputs[i])->set(values[i]);
    work_item<T> item(inputs, outputs, setters);

    std::vector<std::vector<T>> columns = values;
    std::vector<std::vector<T>> expected(outputs.size(), std::vector<T> (n));
    std::vector<T *> column_pointers, output_pointers;
    for (auto &c : columns) column_pointers.push_back(c.data());
    for (auto &o : expected) output_pointers.push_back(o.data());
    for (int l = 0; l < launches; l++) item.run(n, column_pointers, output_pointers);

    graph::input_nodes<T> in;
    graph::map_nodes<T> set;
    to_lists<T> (item, in, set);
    gpu::hip_context<T> gpu(0);
    std::ostringstream source;
    jit::register_map registers;
    gpu.create_header(source);
    add_kernel<T> (gpu, source, registers, "test_kernel", in, item.out_nodes, set, n);
    gpu.compile(source.str(), {"test_kernel"}, false);
    jit::texture1d_list tex1d;
    jit::texture2d_list tex2d;
    auto call = gpu.create_kernel_call("test_kernel", in, item.out_nodes, graph::shared_random_state<T> (), n, tex1d, tex2d);
    for (int l = 0; l < launches; l++) call();
    gpu.wait();

    bool ok = true;
    auto compare = [&] (const T device, const T reference, const char *what, const size_t index, const size_t ray) {
        const bool match = tolerance == static_cast<T> (0) ? same_bits(device, reference)
                         : std::abs(device - reference) <= tolerance*std::max(std::abs(reference), static_cast<T> (1));
        if (!match) {
            ok = false;
            printf("    %s %zu ray %zu: device %.17g reference %.17g\n", what, index, ray,
                   static_cast<double> (device), static_cast<double> (reference));
        }
    };
    std::vector<T> host(n);
    for (size_t o = 0; o < outputs.size(); o++) {
        gpu.copy_to_host(outputs[o], host.data());
        for (size_t r = 0; r < n; r++) compare(host[r], expected[o][r], "output", o, r);
    }
    for (size_t i = 0; i < inputs.size(); i++) {
        gpu.copy_to_host(inputs[i], host.data());
        for (size_t r = 0; r < n; r++) compare(host[r], columns[i][r], "input", i, r);
    }
    cases++;
    if (!ok) failures++;
    printf("  %-58s %s\n", name, ok ? "ok" : "MISMATCH");
}

template<typename T>
std::vector<T> list(std::initializer_list<double> v) {
    std::vector<T> out;
    for (double x : v) out.push_back(static_cast<T> (x));
    return out;
}

template<typename T>
void run_type(const char *label, const T loose) {
    printf("%s\n", label);
    auto a = graph::variable<T> (1, "a");
    auto b = graph::variable<T> (1, "b");
    auto c = graph::variable<T> (1, "c");

//  1-D gathers: arguments below, inside and above the table; scale and offset.
    const std::vector<T> args = list<T> ({-1.5, 0.0, 0.5, 0.999, 1.0, 1.5, 2.5, 3.0, 7.25, -0.0});
    auto p1 = graph::piecewise_1D<T> (list<T> ({1.0, 2.0, 3.0}), a, 1.0, 0.0);
    auto p2 = graph::piecewise_1D<T> (list<T> ({2.0, 4.0, 6.0}), b, 1.0, 0.0);
    auto p3 = graph::piecewise_1D<T> (list<T> ({2.0, 4.0, 6.0}), a, 1.0, 0.0);
    auto p4 = graph::piecewise_1D<T> (list<T> ({1.0, 2.0, 3.0, 5.0, 8.0}), a, 0.5, -1.0);
    const std::vector<T> other = list<T> ({2.5, 1.5, 0.5, -4.0, 0.0, 1.0, 2.0, 2.999, 1.25, 0.75});
    run_case<T> ("piecewise_1D clamp", {a}, {args}, {p1}, {});
    run_case<T> ("piecewise_1D scale 0.5 offset -1", {a}, {args}, {p4}, {});
    run_case<T> ("p1 + p3, p1 - p3 (same argument: tables folded)", {a}, {args}, {p1 + p3, p1 - p3}, {});
    run_case<T> ("p1*p3, p1/p3", {a}, {args}, {p1*p3, p1/p3}, {});
    run_case<T> ("p1*2, p1 + 2, 2/p1, p1 - 0.25", {a}, {args}, {p1*2.0, p1 + 2.0, 2.0/p1, p1 - 0.25}, {});
    run_case<T> ("p1*p2, p1 + p2 (different arguments: not folded)", {a, b}, {args, other}, {p1*p2, p1 + p2, p1/p2}, {});
    run_case<T> ("fma(p1, p3, p2)", {a, b}, {args, other}, {graph::fma(p1, p3, p2)}, {});
    run_case<T> ("fma(p1, a, p3), fma(a, b, p1)", {a, b}, {args, other}, {graph::fma(p1, a, p3), graph::fma(a, b, p1)}, {});
    run_case<T> ("sqrt(p1), p1^2, p1^3", {a}, {args}, {graph::sqrt(p1), graph::pow(p1, 2.0), graph::pow(p1, 3.0)}, {});
    run_case<T> ("pow(p1, p3), exp(p1), log(p1)", {a}, {args}, {graph::pow(p1, p3), graph::exp(p1), graph::log(p1)}, {}, loose);
    run_case<T> ("sin(p1), cos(p1), atan(p1, p3), atan(p1, p2)", {a, b}, {args, other},
                 {graph::sin(p1), graph::cos(p1), graph::atan(p1, p3), graph::atan(p1, p2)}, {}, loose);

//  2-D gathers, row major with num_columns = 2 and 3; 2-D combined with 1-D along rows/columns.
    const std::vector<T> xs = list<T> ({0.5, 0.5, 1.5, 1.5, -3.0, 9.0, 1.0, 0.0, 2.0, 0.25});
    const std::vector<T> ys = list<T> ({0.5, 1.5, 0.5, 1.5, 9.0, -3.0, 1.0, 2.0, 0.0, 2.75});
    auto q1 = graph::piecewise_2D<T> (list<T> ({1.0, 2.0, 3.0, 4.0}), 2, a, 1.0, 0.0, b, 1.0, 0.0);
    auto q3 = graph::piecewise_2D<T> (list<T> ({2.0, 4.0, 6.0, 10.0}), 2, a, 1.0, 0.0, b, 1.0, 0.0);
    auto q6 = graph::piecewise_2D<T> (list<T> ({1.0, 2.0, 3.0, 4.0, 5.0, 6.0}), 3, a, 1.0, 0.0, b, 1.0, 0.0);
    auto q7 = graph::piecewise_2D<T> (list<T> ({1.0, 2.0, 3.0, 4.0, 5.0, 6.0}), 2, a, 0.75, -0.5, b, 2.0, 0.5);
    auto row = graph::piecewise_1D<T> (list<T> ({2.0, 4.0}), a, 1.0, 0.0);
    auto column = graph::piecewise_1D<T> (list<T> ({2.0, 4.0}), b, 1.0, 0.0);
    run_case<T> ("piecewise_2D clamp, 2 and 3 columns", {a, b}, {xs, ys}, {q1, q6, q7}, {});
    run_case<T> ("q1 + q3, q1 - q3, q1*q3, q1/q3", {a, b}, {xs, ys}, {q1 + q3, q1 - q3, q1*q3, q1/q3}, {});
    run_case<T> ("q1*row, q1 + row (combined along rows)", {a, b}, {xs, ys}, {q1*row, q1 + row, q1/row}, {});
    run_case<T> ("q1*column, q1 - column (combined along columns)", {a, b}, {xs, ys}, {q1*column, q1 - column, column/q1}, {});
    run_case<T> ("fma(q1, q3, row), fma(q1, a, b)", {a, b}, {xs, ys}, {graph::fma(q1, q3, row), graph::fma(q1, a, b)}, {});
    run_case<T> ("pow(q1, q3), atan(q1, q3)", {a, b}, {xs, ys}, {graph::pow(q1, q3), graph::atan(q1, q3)}, {}, loose);

//  Arithmetic and math nodes on variables.
    const std::vector<T> u = list<T> ({0.5, 1.5, -2.25, 3.0, 1.0e-3, 7.0, -0.5, 2.0, 10.0, 0.125});
    const std::vector<T> v = list<T> ({2.0, -0.75, 1.5, 0.25, 4.0, -3.0, 8.0, 0.5, 1.0e3, 6.0});
    const std::vector<T> w = list<T> ({1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0, 10.0});
    run_case<T> ("a + b, a - b, a*b, a/b", {a, b}, {u, v}, {a + b, a - b, a*b, a/b}, {});
    run_case<T> ("fma(a, b, c), (a + b)*c - a/c", {a, b, c}, {u, v, w}, {graph::fma(a, b, c), (a + b)*c - a/c}, {});
//  Work items the reference accepts and binds once per node (cuda_context.hpp:330-383,
//  cpu_context.hpp:551-553): an output that is itself an input variable, the same output twice.
    run_case<T> ("outputs {a, a*b, a*b, a + b}: a variable and a repeated output", {a, b}, {u, v}, {a, a*b, a*b, a + b}, {});
    run_case<T> ("a^2, a^3, a^5, sqrt(a*a + 1)", {a}, {u}, {graph::pow(a, 2.0), graph::pow(a, 3.0), graph::pow(a, 5.0), graph::sqrt(a*a + 1.0)}, {});
    run_case<T> ("many quotients of one denominator", {a, b, c}, {u, v, w},
                 {a/(c*c + 1.0) + b/(c*c + 1.0) + (a*b)/(c*c + 1.0), (a - b)/(c*c + 1.0), c/(a*a + b*b)}, {});
    run_case<T> ("exp, log, sin, cos, atan, pow(c, 1.5), pow(c, a)", {a, c}, {u, w},
                 {graph::exp(a), graph::log(c), graph::sin(a), graph::cos(a), graph::atan(c, a), graph::pow(c, 1.5),
                  graph::pow(c, a)}, {}, loose);

//  Setters read the old values; outputs too; repeated launches advance the state.
    run_case<T> ("x <- x + 1, y <- x*y, out = old x - old y, three launches", {a, b}, {u, v}, {a - b},
                 {{a + 1.0, a}, {a*b, b}}, 0, 3);
    run_case<T> ("swap through setters, two launches", {a, b}, {u, v}, {a*b}, {{b, a}, {a, b}}, 0, 2);
    run_case<T> ("gather argument updated by a setter, three launches", {a, b}, {args, other}, {p1 + p2},
                 {{a + 0.5, a}, {b - 0.25, b}}, 0, 3);
}

}  // namespace

int main() {
    run_type<double> ("double", 1.0E-14);
    run_type<float> ("float", 2.0E-6f);
    printf("%d of %d cases identical to the tape\n", cases - failures, cases);
    printf(failures ? "FAIL\n" : "PASS\n");
    return failures ? 1 : 0;
}

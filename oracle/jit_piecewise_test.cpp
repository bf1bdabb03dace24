// ---------------------------------------------------------------------------
// jit_piecewise_test.cpp — TEST INFRASTRUCTURE.  The reference's own
// graph_tests/piecewise_test.cpp (where it lies, unmodified: piecewise_1D :80, piecewise_2D :319,
// index_1D :834, index_2D :868, each through jit::context::add_kernel / compile /
// create_kernel_call / copy_to_host) over gpu::hip_context, real-typed flavours, on the GPU.
// ---------------------------------------------------------------------------
#define main reference_main
#include REFERENCE_TEST
#undef main

int main() {
    run_tests<float> ();
    run_tests<double> ();
    std::cout << "piecewise_test.cpp (float, double) on hip_context: PASS" << std::endl;
    return 0;
}

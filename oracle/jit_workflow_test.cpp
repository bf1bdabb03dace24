// ---------------------------------------------------------------------------
// jit_workflow_test.cpp — TEST INFRASTRUCTURE.  Compiles the reference's own
// graph_tests/workflow_test.cpp (where it lies, unmodified) against the real jit::context /
// workflow::manager over gpu::hip_context and runs its real-typed flavours on the GPU.
//
// The test includes graph_framework.hpp, which pulls in the NetCDF- and LLVM-dependent headers
// (equilibrium.hpp, output.hpp, cpu_context.hpp); its guard is pre-defined and the headers the
// test really uses are included instead.  The reference's main() also runs the complex
// flavours; it is compiled (so hip_context has to compile for them) but not called here.
// ---------------------------------------------------------------------------
#define graph_framework_h
#include "workflow.hpp"
#include "arithmetic.hpp"
#include "math.hpp"
#include "trigonometry.hpp"
#include "piecewise.hpp"

#define main reference_main
#include REFERENCE_TEST
#undef main

int main() {
    run_tests<float> ();
    run_tests<double> ();
    std::cout << "workflow_test.cpp (float, double) on hip_context: PASS" << std::endl;
    return 0;
}

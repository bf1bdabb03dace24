// ---------------------------------------------------------------------------
// jit_hip_prelude.hpp — TEST INFRASTRUCTURE (oracle/Makefile, targets _ref/jit_*).
//
// The reference's jit.hpp is compiled here as the maintainer's patch of INTEGRATION.md leaves
// it (build-directory copies made by sed, never committed), with -DUSE_HIP selecting
// gpu::hip_context.  jit.hpp includes cpu_context.hpp unconditionally; that header needs the
// Clang/LLVM ORC development headers (cpu_context.hpp:21-38), which this image lacks, so its
// include guard is pre-defined (-Dcpu_context_h) and the one thing jit.hpp still names from it
// — the class template, as the branch of std::conditional that USE_HIP never selects
// (jit.hpp:63-71) — is declared, not defined.
// ---------------------------------------------------------------------------
#ifndef jit_hip_prelude_hpp
#define jit_hip_prelude_hpp

namespace gpu {
    template<jit::float_scalar T, bool SAFE_MATH> class cpu_context;
}

#endif /* jit_hip_prelude_hpp */

/* ---------------------------------------------------------------------------
 * gfir.h — GFIR: the serialized form of one graph_framework work item.
 *
 * A work item is what the reference hands to its backend context for ONE fused
 * kernel: input variables, output expressions and setters (expression ->
 * variable), i.e. the arguments of jit::context::add_kernel
 * (graph_framework/jit.hpp:118-194) as assembled by workflow::work_item
 * (graph_framework/workflow.hpp:22-76).  GFIR stores the expression DAG itself
 * (one record per hash-consed node, in the order leaf_node::compile() recurses:
 * left, [middle,] right, self — e.g. arithmetic.hpp:645-669, :5079-5127), not
 * the C++ text the reference generates from it, so that a backend can lower
 * it without a C++ front end.
 *
 * Producers: graph_framework_amd/gfir_serialize.hpp (walks a reference DAG
 * through the public *_cast API; used in-process by hip_context.hpp and by the
 * exporter that writes the committed workload files).
 * Consumers: graph_framework_amd/csrc (HIP lowering) and oracle/ (CPU
 * interpreter).
 *
 * Layout (little endian, everything 4-byte aligned):
 *   gfir_header
 *   char        name[name_bytes]                  (name_bytes % 4 == 0, NUL padded)
 *   per input:  uint32 bytes; char symbol[bytes]  (bytes % 4 == 0, NUL padded)
 *   per table:  gfir_table_header; double data[rows*cols]   (values exactly
 *               representable in the item's dtype; complex items: (re, im) pairs,
 *               2*rows*cols doubles)
 *   gfir_instruction ins[num_instructions]
 *   uint32      outputs[num_outputs]              (instruction index)
 *   gfir_setter setters[num_setters]
 * ------------------------------------------------------------------------- */
#ifndef GFIR_H
#define GFIR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFIR_MAGIC "GFIR0001"

enum gfir_dtype {
    GFIR_F32 = 0,   /* graph_type FLOAT  (graph_c_binding.h:153-158) */
    GFIR_F64 = 1,   /* graph_type DOUBLE */
    GFIR_C32 = 2,   /* graph_type COMPLEX_FLOAT:  elements are (re, im) pairs of float  */
    GFIR_C64 = 3    /* graph_type COMPLEX_DOUBLE: elements are (re, im) pairs of double */
};

/* gfir_header.flags */
#define GFIR_SAFE_MATH 1u   /* the item was built with SAFE_MATH = true: guarded multiply, divide, fma and exp
                               (arithmetic.hpp:2534-2557, :3526-3541, :5101-5117, math.hpp:450-471) and
                               NaN -> 0 on every store (cpu_context.hpp:530-547) */

/* One value per reference node class that can appear in a real-valued kernel. */
enum gfir_op {
    GFIR_CONST   = 0,   /* constant_node        node.hpp:729      imm[0] = value (complex items: imm[0] + i imm[1]) */
    GFIR_INPUT   = 1,   /* variable_node        node.hpp:1386     a = input index           */
    GFIR_ADD     = 2,   /* add_node             arithmetic.hpp:132   a + b                  */
    GFIR_SUB     = 3,   /* subtract_node        arithmetic.hpp:879   a - b                  */
    GFIR_MUL     = 4,   /* multiply_node        arithmetic.hpp:1720  a*b                    */
    GFIR_DIV     = 5,   /* divide_node          arithmetic.hpp:2769  a/b                    */
    GFIR_FMA     = 6,   /* fma_node             arithmetic.hpp:3736  fma(a, b, c)           */
    GFIR_SQRT    = 7,   /* sqrt_node            math.hpp:26          sqrt(a)                */
    GFIR_POWI    = 8,   /* pow_node, integer exponent  math.hpp:1218-1223  a*a*...*a, aux = exponent */
    GFIR_POW     = 9,   /* pow_node             math.hpp:844         pow(a, b)              */
    GFIR_SIN     = 10,  /* sine_node            trigonometry.hpp:25                         */
    GFIR_COS     = 11,  /* cosine_node          trigonometry.hpp:276                        */
    GFIR_ATAN2   = 12,  /* arctan_node          trigonometry.hpp:553, emits atan2(b, a) :718 */
    GFIR_EXP     = 13,  /* exp_node             math.hpp:337                                */
    GFIR_LOG     = 14,  /* log_node             math.hpp:602                                */
    GFIR_GATHER1 = 15,  /* piecewise_1D_node    piecewise.hpp:105  table[idx(a; imm0 scale, imm1 offset)] */
    GFIR_GATHER2 = 16,  /* piecewise_2D_node    piecewise.hpp:686  table[idx(a; imm0, imm1)*cols + idx(b; imm2, imm3)] */
    GFIR_INDEX1  = 17,  /* index_1D_node        piecewise.hpp:1448 buffer of input c, [idx(a; imm0 scale, imm1 offset)], aux = its length */
    GFIR_INDEX2  = 18,  /* index_2D_node        piecewise.hpp:1788 buffer of input c, [idx(a; imm0, imm1)*aux + idx(b; imm2, imm3)],
                           aux = columns, reserved = rows */
    GFIR_ERFI    = 20,  /* erfi_node            math.hpp:1440      erfi(a), complex items (special::erfi, special_functions.hpp:1583) */
    GFIR_RANDOM  = 19   /* random_node          random.hpp:296     one draw of the kernel's MT19937 state (random.hpp:318-339),
                           converted to the item's type.  The reference prints this node as the TEXT `random(state)`
                           wherever it is used (random.hpp:418), so every use is a draw of its own: one record per use,
                           a = the previous draw (or a constant record for the first), which orders the draws. */
};
/* idx(x; scale, offset) = (uint)min(max((x - offset)/scale, 0), length - 1),
 * compile_index, piecewise.hpp:26-65 (complex items: the real part of the quotient, :42-54). */

struct gfir_header {
    char     magic[8];
    uint32_t dtype;             /* enum gfir_dtype */
    uint32_t num_inputs;
    uint32_t num_outputs;
    uint32_t num_setters;
    uint32_t num_tables;
    uint32_t num_instructions;
    uint32_t name_bytes;
    uint32_t flags;             /* GFIR_SAFE_MATH */
};

struct gfir_table_header {
    uint32_t rows;              /* 1 for piecewise_1D */
    uint32_t cols;
};

struct gfir_instruction {
    uint32_t op;                /* enum gfir_op */
    uint32_t a, b, c;           /* operand instruction indices (unused = 0xFFFFFFFF) */
    uint32_t aux;               /* POWI exponent; GATHER table index; INDEX1 length; INDEX2 columns */
    uint32_t reserved;          /* INDEX2 rows; otherwise 0 */
    double   imm[4];
};

struct gfir_setter {
    uint32_t value;             /* instruction index of the expression */
    uint32_t input;             /* index of the variable it overwrites */
};

#define GFIR_NONE 0xFFFFFFFFu

#ifdef __cplusplus
}
#endif

#endif /* GFIR_H */

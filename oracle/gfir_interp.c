/* ---------------------------------------------------------------------------
 * gfir_interp.c — CPU ORACLE for GFIR work items (test infrastructure only).
 *
 * Executes a serialized graph_framework work item (include/gfir.h) one record
 * at a time in strict IEEE arithmetic: exactly one operation per reference
 * node, in the arithmetic the node's compile() method emits —
 *   add/sub/mul/div  arithmetic.hpp:645-669, :1475, :2516, :3508
 *   fma              arithmetic.hpp:5079-5127 (one real fma)
 *   sqrt             math.hpp:166
 *   pow              math.hpp:1199-1230 (integer exponent = repeated multiply)
 *   gathers          piecewise.hpp:26-65 (index clamp), :349-437, :1072-1208
 * followed by the stores of cpu_context::create_kernel_postfix
 * (cpu_context.hpp:522-580): setters first, then outputs.  It is the CPU
 * restatement of gpu::cpu_context's serial kernel loop (cpu_context.hpp:487).
 *
 * Pinned bit-for-bit against oracle/_ref/gf_ref (the reference's own graph
 * layer) and through it against SURVEY.md §8(c)'s probe values and
 * graph_tests/efit_gold.nc; see tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load
 * this.  Built with -ffp-contract=off: no fma except where the graph has one.
 * ------------------------------------------------------------------------- */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>

#include "../include/gfir.h"

typedef struct {
    struct gfir_header header;
    char *name;
    struct gfir_instruction *ins;
    uint32_t *outputs;
    struct gfir_setter *setters;
    uint32_t *table_rows, *table_cols;
    double **tables_f64;
    float **tables_f32;
} gfi_item;

#define TAKE(dst, bytes) do { if (pos + (bytes) > len) { gfi_free(item); return NULL; } \
                              memcpy((dst), data + pos, (bytes)); pos += (bytes); } while (0)

void gfi_free(gfi_item *item) {
    if (!item) return;
    if (item->tables_f64) for (uint32_t i = 0; i < item->header.num_tables; i++) free(item->tables_f64[i]);
    if (item->tables_f32) for (uint32_t i = 0; i < item->header.num_tables; i++) free(item->tables_f32[i]);
    free(item->tables_f64); free(item->tables_f32);
    free(item->table_rows); free(item->table_cols);
    free(item->name); free(item->ins); free(item->outputs); free(item->setters);
    free(item);
}

gfi_item *gfi_load(const uint8_t *data, const size_t len) {
    size_t pos = 0;
    gfi_item *item = calloc(1, sizeof(gfi_item));
    TAKE(&item->header, sizeof(struct gfir_header));
    if (memcmp(item->header.magic, GFIR_MAGIC, 8) != 0) { gfi_free(item); return NULL; }
    const struct gfir_header *h = &item->header;
    item->name = calloc(h->name_bytes + 1, 1);
    TAKE(item->name, h->name_bytes);
    for (uint32_t i = 0; i < h->num_inputs; i++) {
        uint32_t bytes;
        TAKE(&bytes, 4);
        if (pos + bytes > len) { gfi_free(item); return NULL; }
        pos += bytes;
    }
    item->table_rows = calloc(h->num_tables + 1, sizeof(uint32_t));
    item->table_cols = calloc(h->num_tables + 1, sizeof(uint32_t));
    item->tables_f64 = calloc(h->num_tables + 1, sizeof(double *));
    item->tables_f32 = calloc(h->num_tables + 1, sizeof(float *));
    for (uint32_t i = 0; i < h->num_tables; i++) {
        struct gfir_table_header th;
        TAKE(&th, sizeof(th));
        const size_t count = (size_t)th.rows*th.cols*(h->dtype >= GFIR_C32 ? 2 : 1);
        item->table_rows[i] = th.rows;
        item->table_cols[i] = th.cols;
        item->tables_f64[i] = malloc(count*sizeof(double));
        item->tables_f32[i] = malloc(count*sizeof(float));
        TAKE(item->tables_f64[i], count*sizeof(double));
        for (size_t k = 0; k < count; k++) item->tables_f32[i][k] = (float)item->tables_f64[i][k];
    }
    item->ins = malloc(sizeof(struct gfir_instruction)*(h->num_instructions + 1));
    TAKE(item->ins, sizeof(struct gfir_instruction)*h->num_instructions);
    item->outputs = malloc(sizeof(uint32_t)*(h->num_outputs + 1));
    TAKE(item->outputs, sizeof(uint32_t)*h->num_outputs);
    item->setters = malloc(sizeof(struct gfir_setter)*(h->num_setters + 1));
    TAKE(item->setters, sizeof(struct gfir_setter)*h->num_setters);
    return item;
}

void gfi_info(const gfi_item *item, uint32_t *info6) {
    info6[0] = item->header.dtype;
    info6[1] = item->header.num_inputs;
    info6[2] = item->header.num_outputs;
    info6[3] = item->header.num_setters;
    info6[4] = item->header.num_tables;
    info6[5] = item->header.num_instructions;
}

/* header flags (GFIR_SAFE_MATH) | 0x100 if the item draws random numbers. */
uint32_t gfi_flags(const gfi_item *item) {
    uint32_t flags = item->header.flags;
    for (uint32_t i = 0; i < item->header.num_instructions; i++) {
        if (item->ins[i].op == GFIR_RANDOM) flags |= 0x100u;
    }
    return flags;
}

/* Which inputs a setter overwrites (info for callers). */
void gfi_setter_inputs(const gfi_item *item, uint32_t *out) {
    for (uint32_t i = 0; i < item->header.num_setters; i++) out[i] = item->setters[i].input;
}

#define DEFINE_RUN(SUFFIX, REAL, TABLES, FMA, SQRT, POW, SIN, COS, ATAN2, EXP, LOG, FMIN, FMAX)          \
static inline size_t gfi_index_##SUFFIX(const REAL x, const REAL scale, const REAL offset,               \
                                        const uint32_t length) {                                         \
    const REAL q = (x - offset)/scale;                                                                   \
    return (size_t)FMIN(FMAX(q, (REAL)0), (REAL)(length - 1));                                           \
}                                                                                                        \
/* Elements [begin, end) of SoA columns; outs[o][i] receive outputs, setters update columns in place. */ \
void gfi_run_##SUFFIX(const gfi_item *item, REAL **columns, REAL **outs,                                 \
                      const size_t begin, const size_t end) {                                            \
    const uint32_t n_ins = item->header.num_instructions;                                                \
    REAL *r = malloc(sizeof(REAL)*(n_ins + 1));                                                          \
    for (size_t e = begin; e < end; e++) {                                                               \
        for (uint32_t i = 0; i < n_ins; i++) {                                                           \
            const struct gfir_instruction *c = &item->ins[i];                                            \
            switch (c->op) {                                                                             \
                case GFIR_CONST: r[i] = (REAL)c->imm[0]; break;                                          \
                case GFIR_INPUT: r[i] = columns[c->a][e]; break;                                         \
                case GFIR_ADD:   r[i] = r[c->a] + r[c->b]; break;                                        \
                case GFIR_SUB:   r[i] = r[c->a] - r[c->b]; break;                                        \
                case GFIR_MUL:   r[i] = r[c->a]*r[c->b]; break;                                          \
                case GFIR_DIV:   r[i] = r[c->a]/r[c->b]; break;                                          \
                case GFIR_FMA:   r[i] = FMA(r[c->a], r[c->b], r[c->c]); break;                           \
                case GFIR_SQRT:  r[i] = SQRT(r[c->a]); break;                                            \
                case GFIR_POWI: {                                                                        \
                    REAL v = r[c->a];                                                                    \
                    for (uint32_t k = 1; k < c->aux; k++) v = v*r[c->a];                                 \
                    r[i] = v;                                                                            \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_POW:   r[i] = POW(r[c->a], r[c->b]); break;                                    \
                case GFIR_SIN:   r[i] = SIN(r[c->a]); break;                                             \
                case GFIR_COS:   r[i] = COS(r[c->a]); break;                                             \
                case GFIR_ATAN2: r[i] = ATAN2(r[c->b], r[c->a]); break;                                  \
                case GFIR_EXP:   r[i] = EXP(r[c->a]); break;                                             \
                case GFIR_LOG:   r[i] = LOG(r[c->a]); break;                                             \
                case GFIR_GATHER1:                                                                       \
                    r[i] = item->TABLES[c->aux][gfi_index_##SUFFIX(r[c->a], (REAL)c->imm[0],             \
                                                                   (REAL)c->imm[1],                      \
                                                                   item->table_cols[c->aux])];           \
                    break;                                                                               \
                case GFIR_GATHER2:                                                                       \
                    r[i] = item->TABLES[c->aux][gfi_index_##SUFFIX(r[c->a], (REAL)c->imm[0],             \
                                                                   (REAL)c->imm[1],                      \
                                                                   item->table_rows[c->aux])             \
                                                *item->table_cols[c->aux] +                              \
                                                gfi_index_##SUFFIX(r[c->b], (REAL)c->imm[2],             \
                                                                   (REAL)c->imm[3],                      \
                                                                   item->table_cols[c->aux])];           \
                    break;                                                                               \
                case GFIR_INDEX1:                                                                            \
                    r[i] = columns[c->c][gfi_index_##SUFFIX(r[c->a], (REAL)c->imm[0], (REAL)c->imm[1], c->aux)];    \
                    break;                                                                                   \
                case GFIR_INDEX2:                                                                            \
                    r[i] = columns[c->c][gfi_index_##SUFFIX(r[c->a], (REAL)c->imm[0], (REAL)c->imm[1], c->reserved) \
                                         *c->aux +                                                           \
                                         gfi_index_##SUFFIX(r[c->b], (REAL)c->imm[2], (REAL)c->imm[3], c->aux)];    \
                    break;                                                                                   \
                default: r[i] = (REAL)NAN;                                                               \
            }                                                                                            \
        }                                                                                                \
        for (uint32_t s = 0; s < item->header.num_setters; s++) {                                        \
            columns[item->setters[s].input][e] = r[item->setters[s].value];                              \
        }                                                                                                \
        for (uint32_t o = 0; o < item->header.num_outputs; o++) {                                        \
            outs[o][e] = r[item->outputs[o]];                                                            \
        }                                                                                                \
    }                                                                                                    \
    free(r);                                                                                             \
}                                                                                                        \
typedef struct {                                                                                         \
    const gfi_item *item; REAL **columns; REAL **outs; size_t begin, end, steps;                         \
} gfi_job_##SUFFIX;                                                                                      \
static void *gfi_worker_##SUFFIX(void *p) {                                                              \
    gfi_job_##SUFFIX *j = p;                                                                             \
    for (size_t s = 0; s < j->steps; s++) gfi_run_##SUFFIX(j->item, j->columns, j->outs, j->begin, j->end); \
    return NULL;                                                                                         \
}                                                                                                        \
/* `steps` passes over n elements on `threads` host threads, contiguous shards as the reference splits  \
   them (graph_benchmark/xrays_bench.cpp:38-51).  Returns wall seconds. */                              \
double gfi_run_threads_##SUFFIX(const gfi_item *item, REAL **columns, REAL **outs, const size_t n,       \
                                const size_t steps, size_t threads) {                                    \
    if (threads < 1) threads = 1;                                                                        \
    if (threads > n) threads = n ? n : 1;                                                                \
    pthread_t *ids = malloc(sizeof(pthread_t)*threads);                                                  \
    gfi_job_##SUFFIX *jobs = malloc(sizeof(gfi_job_##SUFFIX)*threads);                                   \
    const size_t batch = n/threads, extra = n%threads;                                                   \
    struct timespec t0, t1;                                                                              \
    clock_gettime(CLOCK_MONOTONIC, &t0);                                                                 \
    size_t begin = 0;                                                                                    \
    for (size_t t = 0; t < threads; t++) {                                                               \
        const size_t count = batch + (extra > t ? 1 : 0);                                                \
        jobs[t] = (gfi_job_##SUFFIX){item, columns, outs, begin, begin + count, steps};                  \
        begin += count;                                                                                  \
        pthread_create(&ids[t], NULL, gfi_worker_##SUFFIX, &jobs[t]);                                    \
    }                                                                                                    \
    for (size_t t = 0; t < threads; t++) pthread_join(ids[t], NULL);                                     \
    clock_gettime(CLOCK_MONOTONIC, &t1);                                                                 \
    free(ids); free(jobs);                                                                               \
    return (double)(t1.tv_sec - t0.tv_sec) + 1.0e-9*(double)(t1.tv_nsec - t0.tv_nsec);                   \
}

DEFINE_RUN(f64, double, tables_f64, fma, sqrt, pow, sin, cos, atan2, exp, log, fmin, fmax)
DEFINE_RUN(f32, float, tables_f32, fmaf, sqrtf, powf, sinf, cosf, atan2f, expf, logf, fminf, fmaxf)


/* ---------------------------------------------------------------------------
 * Items off the hot path: complex base types (GFIR_C32/GFIR_C64), SAFE_MATH guards, random draws.
 * One value = (re, im) in the item's base precision; real items keep im = 0 and use the real
 * formulas.  Complex arithmetic as graph_framework_amd/csrc/prelude.hpp states it (textbook
 * product, Smith's quotient, every operation unfused): no fixture of the reference pins complex
 * kernels, so this is the restatement both sides are held to ("parity unpinned").
 * SAFE_MATH: arithmetic.hpp:2534-2557, :3526-3541, :5101-5117, math.hpp:450-471,
 * cpu_context.hpp:530-547.  Random: random.hpp:318-339, one MT19937 state per lane of 1024
 * (cuda_context.hpp:509-522, :817): element e draws from state e % 1024, elements in order.
 * ------------------------------------------------------------------------- */
typedef struct { uint32_t array[624]; uint16_t index; } gfi_mt_state;

static uint32_t gfi_random(gfi_mt_state *state) {
    const uint16_t k = state->index;
    uint16_t j = (uint16_t)((k + 1)%624);
    uint32_t x = (state->array[k] & 0x80000000u) | (state->array[j] & 0x7fffffffu);
    uint32_t xa = x >> 1;
    if (x & 1u) xa ^= 0x9908b0dfu;
    j = (uint16_t)((k + 397)%624);
    x = state->array[j]^xa;
    state->array[k] = x;
    state->index = (uint16_t)((k + 1)%624);
    uint32_t y = x^(x >> 11);
    y = y^((y << 7) & 0x9d2c5680u);
    y = y^((y << 15) & 0xefc60000u);
    return y^(y >> 18);
}

/* random_state_node::initialize_state, random.hpp:104-113: `count` states seeded seed, seed + 1, ... */
void gfi_random_states(gfi_mt_state *states, const size_t count, const uint32_t seed) {
    for (size_t s = 0; s < count; s++) {
        states[s].array[0] = seed + (uint32_t)s;
        for (uint32_t i = 1; i < 624; i++) {
            states[s].array[i] = 1812433253u*(states[s].array[i - 1]^(states[s].array[i - 1] >> 30)) + i;
        }
        states[s].index = 0;
    }
}

/* erfi(z) for complex z as graph_framework_amd/csrc/prelude.hpp evaluates it: -i erf(iz),
 * erf(u) = 1 - exp(-u^2) w(iu), w by Weideman's N = 48 rational approximation (SIAM J. Numer. Anal.
 * 31 (1994) 1497).  The reference's special::erfi (special_functions.hpp:1583) is pinned by its own
 * fixture graph_tests/test_erfi.nc at 2e-14 (erfi_test.cpp:20-83); tests/test_oracle.py holds this
 * restatement to the same fixture and tolerance. */
static double gfi_weideman[48], gfi_weideman_l = 0.0;
static void gfi_weideman_init(void) {
    const int n = 48, m = 2*n, m2 = 2*m;
    const long double pi = 3.141592653589793238462643383279502884L;
    const long double l = sqrtl((long double)n/sqrtl(2.0L));
    long double f[192];
    f[0] = 0.0L;
    for (int j = 1; j < m2; j++) {
        const long double t = l*tanl((j - m)*pi/m/2.0L);
        f[j] = expl(-t*t)*(l*l + t*t);
    }
    for (int k = 1; k <= n; k++) {
        long double sum = 0.0L;
        for (int j = 0; j < m2; j++) sum += f[(j + m2/2)%m2]*cosl(2.0L*pi*k*j/m2);
        gfi_weideman[k - 1] = (double)(sum/m2);
    }
    gfi_weideman_l = (double)l;
}
typedef struct { double re, im; } gfi_z;
static inline gfi_z gfi_zmul(const gfi_z a, const gfi_z b) { gfi_z v = {a.re*b.re - a.im*b.im, a.re*b.im + a.im*b.re}; return v; }
static inline gfi_z gfi_zdiv(const gfi_z a, const gfi_z b) {
    const double d = b.re*b.re + b.im*b.im;
    gfi_z v = {(a.re*b.re + a.im*b.im)/d, (a.im*b.re - a.re*b.im)/d};
    return v;
}
static inline gfi_z gfi_zexp(const gfi_z a) { const double e = exp(a.re); gfi_z v = {e*cos(a.im), e*sin(a.im)}; return v; }
static gfi_z gfi_faddeeva_upper(const gfi_z z) {
    const gfi_z down = {gfi_weideman_l + z.im, -z.re}, up = {gfi_weideman_l - z.im, z.re};
    const gfi_z big = gfi_zdiv(up, down);
    gfi_z p = {gfi_weideman[47], 0.0};
    for (int k = 46; k >= 0; k--) {
        p = gfi_zmul(p, big);
        p.re += gfi_weideman[k];
    }
    const gfi_z twice = {2.0*p.re, 2.0*p.im}, scale = {0x1.20dd750429b6dp-1, 0.0};
    const gfi_z first = gfi_zdiv(twice, gfi_zmul(down, down)), second = gfi_zdiv(scale, down);
    gfi_z v = {first.re + second.re, first.im + second.im};
    return v;
}
static gfi_z gfi_faddeeva(const gfi_z z) {
    if (z.im >= 0.0) return gfi_faddeeva_upper(z);
    const gfi_z square = gfi_zmul(z, z), minus = {-square.re, -square.im}, mirrored = {-z.re, -z.im};
    const gfi_z e = gfi_zexp(minus), w = gfi_faddeeva_upper(mirrored);
    gfi_z v = {2.0*e.re - w.re, 2.0*e.im - w.im};
    return v;
}
void gfi_erfi(const double re, const double im, double *out) {
    if (gfi_weideman_l == 0.0) gfi_weideman_init();
/*  The branches special::erf_complex takes before its general formula (special_functions.hpp:1495-1517),
 *  as behaviour: an argument on the imaginary axis gives i erf(Im z); an argument on the real axis gives a
 *  REAL value exp(x^2) Im w(x), saturated at the largest double beyond x^2 = 720 (so that
 *  Im Z(zeta) = sqrt(pi) exp(-zeta^2) keeps its relative accuracy in dispersion.hpp:289-297);
 *  Re(z^2) < -750 gives -+i. */
    if (re == 0.0) { out[0] = re; out[1] = erf(im); return; }
    if (im == 0.0) {
        const gfi_z on_axis = {re, 0.0};
        out[0] = re*re > 720.0 ? copysign(1.7976931348623157e308, re) : exp(re*re)*gfi_faddeeva_upper(on_axis).im;
        out[1] = im;
        return;
    }
    if (isinf(re) && isinf(im)) { out[0] = 0.0; out[1] = -0.0; return; }
    if ((re + im)*(re - im) < -750.0) { out[0] = 0.0; out[1] = im <= 0.0 ? -1.0 : 1.0; return; }
    const gfi_z z = {re, im}, mirrored = {-re, -im};
/*  Small arguments, in the regions where special::erf_complex leaves its general formula
 *  (special_functions.hpp:1534-1553, u = iz): the Maclaurin series of erfi for |Im z| < 0.08, |Re z| < 0.01;
 *  the expansion of erf(x + iy) about the imaginary axis for |Im z| < 0.005, |2 Re z Im z| < 0.005
 *  (formulas: graph_framework_amd/csrc/prelude.hpp, gf_erfi). */
    if (fabs(im) < 8.0e-2) {
        if (fabs(re) < 1.0e-2) {
            static const double c[6] = {0x1.c02db40040b85p-11, 0x1.565bcd0e6a53fp-8, 0x1.b82ce31288b51p-6,
                                        0x1.ce2f21a042be2p-4, 0x1.812746b0379e7p-2, 0x1.20dd750429b6dp+0};
            const gfi_z s = gfi_zmul(z, z);
            gfi_z p = {0x1.f9a326f9b89b7p-14, 0.0};
            for (int k = 0; k < 6; k++) {
                p = gfi_zmul(p, s);
                p.re += c[k];
            }
            const gfi_z v = gfi_zmul(z, p);
            out[0] = v.re;
            out[1] = v.im;
            return;
        }
        if (fabs(im) < 5.0e-3 && fabs(2.0*re*im) < 5.0e-3) {
            const double x = -im, y = re, x2 = x*x, y2 = y*y, e = exp(y2);
            const gfi_z on_axis = {y, 0.0};
            const double wim = gfi_faddeeva_upper(on_axis).im;
            const double erf_re = e*x*(0x1.20dd750429b6dp+0 - x2*(0x1.812746b0379e7p-2 + 0x1.812746b0379e7p-1*y2)
                                       + x2*x2*(0x1.ce2f21a042be2p-4 + y2*(0x1.ce2f21a042be2p-2 + 0x1.341f6bc02c7ecp-3*y2)));
            const double erf_im = e*(wim - x2*y*(0x1.20dd750429b6dp+0 - x2*(0x1.20dd750429b6dp-1 + 0x1.812746b0379e7p-2*y2)));
            out[0] = erf_im;
            out[1] = -erf_re;
            return;
        }
    }
    const gfi_z e = gfi_zexp(gfi_zmul(z, z)), w = gfi_faddeeva(mirrored);
    const double erf_re = 1.0 - (e.re*w.re - e.im*w.im), erf_im = -(e.re*w.im + e.im*w.re);
    out[0] = erf_im;
    out[1] = -erf_re;
}

#define DEFINE_GENERIC(SUFFIX, REAL, TABLES, FMA, SQRT, POW, SIN, COS, ATAN2, EXP, LOG, FMIN, FMAX, FABS, HYPOT, SINH, COSH, BIG) \
typedef struct { REAL re, im; } gfi_value_##SUFFIX;                                                      \
static inline gfi_value_##SUFFIX gfi_mul_##SUFFIX(const gfi_value_##SUFFIX a, const gfi_value_##SUFFIX b, const int cx) { \
    gfi_value_##SUFFIX v = {a.re*b.re, 0};                                                               \
    if (cx) { v.re = a.re*b.re - a.im*b.im; v.im = a.re*b.im + a.im*b.re; }                              \
    return v;                                                                                            \
}                                                                                                        \
static inline gfi_value_##SUFFIX gfi_div_##SUFFIX(const gfi_value_##SUFFIX a, const gfi_value_##SUFFIX b, const int cx) { \
    gfi_value_##SUFFIX v = {0, 0};                                                                       \
    if (!cx) { v.re = a.re/b.re; return v; }                                                             \
    if (FABS(b.re) < FABS(b.im)) {                                                                       \
        const REAL ratio = b.re/b.im, denom = b.re*ratio + b.im;                                         \
        v.re = (a.re*ratio + a.im)/denom; v.im = (a.im*ratio - a.re)/denom;                              \
    } else {                                                                                             \
        const REAL ratio = b.im/b.re, denom = b.im*ratio + b.re;                                         \
        v.re = (a.im*ratio + a.re)/denom; v.im = (a.im - a.re*ratio)/denom;                              \
    }                                                                                                    \
    return v;                                                                                            \
}                                                                                                        \
static inline gfi_value_##SUFFIX gfi_cexp_##SUFFIX(const gfi_value_##SUFFIX a) {                         \
    const REAL e = EXP(a.re);                                                                            \
    gfi_value_##SUFFIX v = {e*COS(a.im), e*SIN(a.im)};                                                   \
    return v;                                                                                            \
}                                                                                                        \
static inline gfi_value_##SUFFIX gfi_clog_##SUFFIX(const gfi_value_##SUFFIX a) {                         \
    gfi_value_##SUFFIX v = {LOG(HYPOT(a.re, a.im)), ATAN2(a.im, a.re)};                                  \
    return v;                                                                                            \
}                                                                                                        \
static inline size_t gfi_gindex_##SUFFIX(const gfi_value_##SUFFIX x, const REAL scale, const REAL offset, \
                                         const uint32_t length, const int cx) {                          \
    const gfi_value_##SUFFIX numerator = {x.re - offset, x.im}, denominator = {scale, 0};                \
    const REAL q = gfi_div_##SUFFIX(numerator, denominator, cx).re;                                      \
    return (size_t)FMIN(FMAX(q, (REAL)0), (REAL)(length - 1));                                           \
}                                                                                                        \
/* Elements [begin, end); columns/outs hold (re, im) pairs for complex items, plain values otherwise. */ \
void gfi_run_generic_##SUFFIX(const gfi_item *item, REAL **columns, REAL **outs, const size_t begin,    \
                              const size_t end, gfi_mt_state *states) {                                  \
    const uint32_t n_ins = item->header.num_instructions;                                                \
    const int cx = item->header.dtype >= GFIR_C32;                                                       \
    const int safe = (item->header.flags & GFIR_SAFE_MATH) != 0;                                         \
    const size_t parts = cx ? 2 : 1;                                                                     \
    gfi_value_##SUFFIX *r = malloc(sizeof(gfi_value_##SUFFIX)*(n_ins + 1));                              \
    const gfi_value_##SUFFIX zero = {0, 0};                                                              \
    for (size_t e = begin; e < end; e++) {                                                               \
        for (uint32_t i = 0; i < n_ins; i++) {                                                           \
            const struct gfir_instruction *c = &item->ins[i];                                            \
            const gfi_value_##SUFFIX a = c->a < n_ins ? r[c->a] : zero, b = c->b < n_ins ? r[c->b] : zero, \
                                     m = c->c < n_ins ? r[c->c] : zero;                                  \
            gfi_value_##SUFFIX v = zero;                                                                 \
            switch (c->op) {                                                                             \
                case GFIR_CONST: v.re = (REAL)c->imm[0]; v.im = cx ? (REAL)c->imm[1] : 0; break;         \
                case GFIR_INPUT: v.re = columns[c->a][e*parts]; v.im = cx ? columns[c->a][e*parts + 1] : 0; break; \
                case GFIR_ADD: v.re = a.re + b.re; v.im = a.im + b.im; break;                            \
                case GFIR_SUB: v.re = a.re - b.re; v.im = a.im - b.im; break;                            \
                case GFIR_MUL:                                                                           \
                    if (safe && ((a.re == 0 && a.im == 0) || (b.re == 0 && b.im == 0))) break;           \
                    v = gfi_mul_##SUFFIX(a, b, cx); break;                                               \
                case GFIR_DIV:                                                                           \
                    if (safe && a.re == 0 && a.im == 0) break;                                           \
                    v = gfi_div_##SUFFIX(a, b, cx); break;                                               \
                case GFIR_FMA:                                                                           \
                    if (safe && ((a.re == 0 && a.im == 0) || (b.re == 0 && b.im == 0))) { v = m; break; } \
                    if (cx) { v = gfi_mul_##SUFFIX(a, b, 1); v.re += m.re; v.im += m.im; }               \
                    else v.re = FMA(a.re, b.re, m.re);                                                   \
                    break;                                                                               \
                case GFIR_SQRT:                                                                          \
                    if (!cx) { v.re = SQRT(a.re); break; }                                               \
                    if (a.re == 0 && a.im == 0) { v.im = a.im; break; }                                  \
                    {   const REAL mod = HYPOT(a.re, a.im);                                              \
                        if (a.re >= 0) { const REAL t = SQRT((mod + a.re)*(REAL)0.5); v.re = t; v.im = a.im/(t + t); } \
                        else { const REAL t = SQRT((mod - a.re)*(REAL)0.5); v.re = FABS(a.im)/(t + t); v.im = a.im < 0 ? -t : t; } \
                    }                                                                                    \
                    break;                                                                               \
                case GFIR_POWI:                                                                          \
                    v = a;                                                                               \
                    for (uint32_t k = 1; k < c->aux; k++) v = gfi_mul_##SUFFIX(v, a, cx);                \
                    break;                                                                               \
                case GFIR_POW:                                                                           \
                    if (!cx) { v.re = POW(a.re, b.re); break; }                                          \
                    if (a.re == 0 && a.im == 0) { v.re = (b.re == 0 && b.im == 0) ? 1 : 0; break; }      \
                    v = gfi_cexp_##SUFFIX(gfi_mul_##SUFFIX(b, gfi_clog_##SUFFIX(a), 1));                 \
                    break;                                                                               \
                case GFIR_SIN:                                                                           \
                    if (cx) { v.re = SIN(a.re)*COSH(a.im); v.im = COS(a.re)*SINH(a.im); } else v.re = SIN(a.re); \
                    break;                                                                               \
                case GFIR_COS:                                                                           \
                    if (cx) { v.re = COS(a.re)*COSH(a.im); v.im = -SIN(a.re)*SINH(a.im); } else v.re = COS(a.re); \
                    break;                                                                               \
                case GFIR_ATAN2:                                                                         \
                    if (!cx) { v.re = ATAN2(b.re, a.re); break; }                                        \
                    {   const gfi_value_##SUFFIX z = gfi_div_##SUFFIX(b, a, 1);                          \
                        const gfi_value_##SUFFIX up = {z.re, (REAL)1 + z.im}, down = {-z.re, (REAL)1 - z.im}; \
                        const gfi_value_##SUFFIX w = gfi_clog_##SUFFIX(gfi_div_##SUFFIX(up, down, 1));   \
                        v.re = -w.im*(REAL)0.5; v.im = w.re*(REAL)0.5;                                   \
                    }                                                                                    \
                    break;                                                                               \
                case GFIR_EXP:                                                                           \
                    if (safe && !(a.re < (REAL)709.8)) { v.re = BIG; break; }                            \
                    if (cx) v = gfi_cexp_##SUFFIX(a); else v.re = EXP(a.re);                             \
                    break;                                                                               \
                case GFIR_LOG:                                                                           \
                    if (cx) v = gfi_clog_##SUFFIX(a); else v.re = LOG(a.re);                             \
                    break;                                                                               \
                case GFIR_ERFI: {                                                                        \
                    double parts2[2];                                                                    \
                    gfi_erfi((double)a.re, (double)a.im, parts2);                                        \
                    v.re = (REAL)parts2[0]; v.im = (REAL)parts2[1];                                      \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_GATHER1: {                                                                     \
                    const size_t at = gfi_gindex_##SUFFIX(a, (REAL)c->imm[0], (REAL)c->imm[1], item->table_cols[c->aux], cx); \
                    v.re = item->TABLES[c->aux][at*parts]; v.im = cx ? item->TABLES[c->aux][at*parts + 1] : 0; \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_GATHER2: {                                                                     \
                    const size_t at = gfi_gindex_##SUFFIX(a, (REAL)c->imm[0], (REAL)c->imm[1], item->table_rows[c->aux], cx) \
                                      *item->table_cols[c->aux] +                                        \
                                      gfi_gindex_##SUFFIX(b, (REAL)c->imm[2], (REAL)c->imm[3], item->table_cols[c->aux], cx); \
                    v.re = item->TABLES[c->aux][at*parts]; v.im = cx ? item->TABLES[c->aux][at*parts + 1] : 0; \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_INDEX1: {                                                                      \
                    const size_t at = gfi_gindex_##SUFFIX(a, (REAL)c->imm[0], (REAL)c->imm[1], c->aux, cx); \
                    v.re = columns[c->c][at*parts]; v.im = cx ? columns[c->c][at*parts + 1] : 0;         \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_INDEX2: {                                                                      \
                    const size_t at = gfi_gindex_##SUFFIX(a, (REAL)c->imm[0], (REAL)c->imm[1], c->reserved, cx)*c->aux + \
                                      gfi_gindex_##SUFFIX(b, (REAL)c->imm[2], (REAL)c->imm[3], c->aux, cx); \
                    v.re = columns[c->c][at*parts]; v.im = cx ? columns[c->c][at*parts + 1] : 0;         \
                    break;                                                                               \
                }                                                                                        \
                case GFIR_RANDOM: v.re = (REAL)gfi_random(&states[e%1024]); break;                       \
                default: v.re = (REAL)NAN;                                                               \
            }                                                                                            \
            r[i] = v;                                                                                    \
        }                                                                                                \
        for (uint32_t s = 0; s < item->header.num_setters; s++) {                                        \
            gfi_value_##SUFFIX v = r[item->setters[s].value];                                            \
            if (safe) { if (v.re != v.re) v.re = 0; if (v.im != v.im) v.im = 0; }                        \
            columns[item->setters[s].input][e*parts] = v.re;                                             \
            if (cx) columns[item->setters[s].input][e*parts + 1] = v.im;                                 \
        }                                                                                                \
        for (uint32_t o = 0; o < item->header.num_outputs; o++) {                                        \
            gfi_value_##SUFFIX v = r[item->outputs[o]];                                                  \
            if (safe) { if (v.re != v.re) v.re = 0; if (v.im != v.im) v.im = 0; }                        \
            outs[o][e*parts] = v.re;                                                                     \
            if (cx) outs[o][e*parts + 1] = v.im;                                                         \
        }                                                                                                \
    }                                                                                                    \
    free(r);                                                                                             \
}

DEFINE_GENERIC(f64, double, tables_f64, fma, sqrt, pow, sin, cos, atan2, exp, log, fmin, fmax, fabs, hypot, sinh, cosh, 1.7976931348623157e308)
DEFINE_GENERIC(f32, float, tables_f32, fmaf, sqrtf, powf, sinf, cosf, atan2f, expf, logf, fminf, fmaxf, fabsf, hypotf, sinhf, coshf, 3.4028234663852886e38f)

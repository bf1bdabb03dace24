// Diagnostic (not product, not a test): which IEEE exception bits does gfx950 accumulate in
// TRAPSTS.EXCP for the operations of the shared-reciprocal division, and can a kernel clear and
// read them?  hipcc --offload-arch=gfx950 -O2 -o trapsts_probe trapsts_probe.hip ; ./trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

// hwreg(HW_REG_TRAPSTS = 3, offset 0, size 9): simm16 = id | offset << 6 | (size - 1) << 11
#define TRAPSTS_EXCP (3 | (0 << 6) | (8 << 11))
#define MODE_ALL     (1 | (0 << 6) | (31 << 11))

__device__ __forceinline__ void clear_flags() { __builtin_amdgcn_s_setreg(TRAPSTS_EXCP, 0u); }
__device__ __forceinline__ unsigned read_flags() { return __builtin_amdgcn_s_getreg(TRAPSTS_EXCP); }

struct result { unsigned flags; unsigned pad; double value; };

__global__ void probe64(const double *a, const double *b, const double *c, result *out, int cases, int only_lane) {
    const int lane = threadIdx.x;
    for (int k = 0; k < cases; k++) {
        const int op = k % 8;
        // lane `only_lane` gets the special operands, the others benign ones (are flags OR-ed over lanes?)
        const bool special = only_lane < 0 || lane == only_lane;
        double x = special ? a[k] : 1.5, y = special ? b[k] : 1.25, z = special ? c[k] : 0.75;
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
        clear_flags();
        double r;
        switch (op) {
            case 0: r = __builtin_fma(x, y, z); break;
            case 1: r = x*y; break;
            case 2: r = x + y; break;
            case 3: r = __builtin_amdgcn_rcp(x); break;
            case 4: r = __builtin_sqrt(x); break;
            case 5: r = x/y; break;                       // the compiler's full sequence
            case 6: { double q = __builtin_amdgcn_rcp(y); double e = __builtin_fma(-y, q, 1.0); q = __builtin_fma(q, e, q);
                      e = __builtin_fma(-y, q, 1.0); q = __builtin_fma(q, e, q);
                      const double p = x*q; const double f = __builtin_fma(-y, p, x); r = __builtin_fma(f, q, p); break; }
            default: r = __builtin_amdgcn_rsq(x); break;
        }
        asm volatile("" : "+v"(r));
        const unsigned flags = read_flags();
        if (lane == (only_lane < 0 ? 0 : only_lane)) { out[k].flags = flags; out[k].value = r; }
        if (lane == 63 && only_lane >= 0) { out[k].pad = flags; }
    }
}

__global__ void probe32(const float *a, const float *b, const float *c, result *out, int cases) {
    for (int k = 0; k < cases; k++) {
        const int op = k % 8;
        float x = a[k], y = b[k], z = c[k];
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
        clear_flags();
        float r;
        switch (op) {
            case 0: r = __builtin_fmaf(x, y, z); break;
            case 1: r = x*y; break;
            case 2: r = x + y; break;
            case 3: r = __builtin_amdgcn_rcpf(x); break;
            case 4: r = __builtin_sqrtf(x); break;
            case 5: r = x/y; break;
            case 6: { float q = __builtin_amdgcn_rcpf(y); float e = __builtin_fmaf(-y, q, 1.0f); q = __builtin_fmaf(e, q, q);
                      const float q0 = x*q; const float e0 = __builtin_fmaf(-y, q0, x); const float q1 = __builtin_fmaf(e0, q, q0);
                      const float e1 = __builtin_fmaf(-y, q1, x); r = __builtin_fmaf(e1, q, q1); break; }
            default: r = __builtin_amdgcn_rsqf(x); break;
        }
        asm volatile("" : "+v"(r));
        const unsigned flags = read_flags();
        if (threadIdx.x == 0) { out[k].flags = flags; out[k].value = r; }
    }
}

__global__ void mode_probe(unsigned *out) {
    out[0] = __builtin_amdgcn_s_getreg(MODE_ALL);
    out[1] = __builtin_amdgcn_s_getreg(3 | (0 << 6) | (31 << 11));
}

static const char *bits(unsigned f) {
    static char buf[128];
    buf[0] = 0;
    const char *names[9] = {"INVALID", "INPUT_DENORM", "DIV0", "OVERFLOW", "UNDERFLOW", "INEXACT", "INT_DIV0", "ADDR_WATCH", "MEM_VIOL"};
    for (int i = 0; i < 9; i++) if (f & (1u << i)) { strcat(buf, names[i]); strcat(buf, " "); }
    return buf;
}

int main() {
    const char *ops[8] = {"fma(x,y,z)", "x*y", "x+y", "rcp(x)", "sqrt(x)", "x/y (compiler)", "x/y (shared-rcp form)", "rsq(x)"};
    struct row { int op; double x, y, z; const char *what; };
    const double inf = INFINITY, den = 4.9406564584124654e-324, tiny = std::ldexp(1.0, -1000), huge = std::ldexp(1.0, 1000);
    std::vector<row> rows = {
        {0, 1.5, 1.25, 0.75, "benign exact"}, {0, 1.0/3.0, 3.0, -1.0, "exact tiny residual (normal)"},
        {0, std::ldexp(1.0/3.0, -530), std::ldexp(3.0, -530), -std::ldexp(1.0, -1060), "residual below 2^-1074: inexact subnormal"},
        {0, std::ldexp(1.0, -540), std::ldexp(1.0, -530), 0.0, "exact subnormal result 2^-1070"},
        {0, inf, 0.0, 1.0, "inf*0"}, {0, huge, huge, 0.0, "overflow"}, {0, den, 1.0, 0.0, "subnormal input, exact"},
        {1, tiny, tiny, 0, "underflow to zero"}, {1, huge, huge, 0, "overflow"}, {1, den, 0.5, 0, "subnormal*0.5 (inexact)"}, {1, 3.0, 0.5, 0, "exact"},
        {2, inf, -inf, 0, "inf-inf"}, {2, 1.0, std::ldexp(1.0, -60), 0, "inexact"}, {2, den, den, 0, "subnormal sum exact"},
        {3, 0.0, 0, 0, "rcp(0)"}, {3, std::ldexp(1.0, 1023), 0, 0, "rcp(2^1023) subnormal"}, {3, 3.0, 0, 0, "rcp(3)"}, {3, inf, 0, 0, "rcp(inf)"}, {3, den, 0, 0, "rcp(denorm)"},
        {4, -1.0, 0, 0, "sqrt(-1)"}, {4, 2.0, 0, 0, "sqrt(2)"},
        {5, 1.0, 3.0, 0, "1/3"}, {5, tiny*std::ldexp(1.0, -40), 3.0, 0, "tiny/3"}, {5, -0.0, 3.0, 0, "-0/3"}, {5, 1.0, 0.0, 0, "1/0"}, {5, den, 3.0, 0, "denorm/3"},
        {6, 1.0, 3.0, 0, "1/3"}, {6, tiny*std::ldexp(1.0, -40), 3.0, 0, "2^-1040/3"}, {6, std::ldexp(1.0, -980), 3.0, 0, "2^-980/3"}, {6, std::ldexp(1.0, -960), 3.0, 0, "2^-960/3"},
        {6, -0.0, 3.0, 0, "-0/3"}, {6, 1.0, 0.0, 0, "1/0"}, {6, den, 3.0, 0, "denorm/3"},
        {6, inf, 3.0, 0, "inf/3"}, {6, 1.0, inf, 0, "1/inf"}, {6, 1.0, std::ldexp(1.0, 600), 0, "1/2^600"}, {6, std::ldexp(1.0, 600), std::ldexp(1.0, -600), 0, "2^600/2^-600 overflow"},
        {6, std::ldexp(1.0, -600), std::ldexp(1.5, 500), 0, "2^-600/1.5*2^500 subnormal quotient"}, {6, 1.0, std::ldexp(1.0, 1023), 0, "1/2^1023"},
        {6, 0.0, 3.0, 0, "0/3"}, {6, 6.0, 3.0, 0, "6/3 exact"},
        {7, 0.0, 0, 0, "rsq(0)"}, {7, 2.0, 0, 0, "rsq(2)"},
    };
    // the kernel picks the op by k % 8: pad rows so that row k has op == k % 8
    std::vector<row> placed;
    for (auto &r : rows) { while (static_cast<int> (placed.size() % 8) != r.op) placed.push_back({static_cast<int> (placed.size() % 8), 1.5, 1.25, 0.75, nullptr}); placed.push_back(r); }
    const int n = placed.size();
    std::vector<double> a(n), b(n), c(n);
    for (int i = 0; i < n; i++) { a[i] = placed[i].x; b[i] = placed[i].y; c[i] = placed[i].z; }
    double *da, *db, *dc; result *dout; unsigned *dmode;
    hipMalloc(&da, n*8); hipMalloc(&db, n*8); hipMalloc(&dc, n*8); hipMalloc(&dout, n*sizeof(result)); hipMalloc(&dmode, 8);
    hipMemcpy(da, a.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n*8, hipMemcpyHostToDevice);
    std::vector<result> out(n);
    mode_probe<<<1, 64>>>(dmode);
    unsigned mode[2]; hipMemcpy(mode, dmode, 8, hipMemcpyDeviceToHost);
    printf("MODE = 0x%08x (fp_round %u, fp_denorm %u, excp_en 0x%x)  TRAPSTS = 0x%08x\n", mode[0], mode[0] & 15, (mode[0] >> 4) & 15, (mode[0] >> 12) & 0x1ff, mode[1]);
    for (int pass = 0; pass < 2; pass++) {
        hipMemset(dout, 0, n*sizeof(result));
        probe64<<<1, 64>>>(da, db, dc, dout, n, pass == 0 ? -1 : 17);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(out.data(), dout, n*sizeof(result), hipMemcpyDeviceToHost);
        printf("---- fp64, %s ----\n", pass == 0 ? "every lane special" : "only lane 17 special (flags read by lane 17 | by lane 63)");
        for (int i = 0; i < n; i++) {
            if (!placed[i].what) continue;
            printf("%-24s %-44s -> %-24.17g flags 0x%03x %s", ops[placed[i].op], placed[i].what, out[i].value, out[i].flags, bits(out[i].flags));
            if (pass == 1) printf(" | 0x%03x", out[i].pad);
            printf("\n");
        }
    }
    // fp32
    {
        struct row32 { int op; float x, y, z; const char *what; };
        const float finf = INFINITY, fden = 1.4e-45f;
        std::vector<row32> r32 = {
            {0, std::ldexp(1.0f/3.0f, -60), std::ldexp(3.0f, -60), -std::ldexp(1.0f, -120), "residual below 2^-149: inexact subnormal"},
            {0, std::ldexp(1.0f, -70), std::ldexp(1.0f, -70), 0.0f, "exact subnormal result 2^-140"}, {0, fden, 1.0f, 0.0f, "subnormal input exact"},
            {1, std::ldexp(1.0f, -100), std::ldexp(1.0f, -100), 0, "underflow to zero"},
            {3, 0.0f, 0, 0, "rcp(0)"}, {3, std::ldexp(1.0f, 127), 0, 0, "rcp(2^127) subnormal"}, {3, 3.0f, 0, 0, "rcp(3)"},
            {6, 1.0f, 3.0f, 0, "1/3"}, {6, std::ldexp(1.0f, -110), 3.0f, 0, "2^-110/3"}, {6, std::ldexp(1.0f, -100), 3.0f, 0, "2^-100/3"}, {6, -0.0f, 3.0f, 0, "-0/3"},
            {6, fden, 3.0f, 0, "denorm/3"}, {6, finf, 3.0f, 0, "inf/3"}, {6, 1.0f, 0.0f, 0, "1/0"}, {6, 1.0f, std::ldexp(1.0f, 110), 0, "1/2^110"}, {6, 6.0f, 3.0f, 0, "6/3"},
            {5, std::ldexp(1.0f, -110), 3.0f, 0, "2^-110/3 compiler"},
        };
        std::vector<row32> p32;
        for (auto &r : r32) { while (static_cast<int> (p32.size() % 8) != r.op) p32.push_back({static_cast<int> (p32.size() % 8), 1.5f, 1.25f, 0.75f, nullptr}); p32.push_back(r); }
        const int m = p32.size();
        std::vector<float> fa(m), fb(m), fc(m);
        for (int i = 0; i < m; i++) { fa[i] = p32[i].x; fb[i] = p32[i].y; fc[i] = p32[i].z; }
        float *xa, *xb, *xc; result *xo;
        hipMalloc(&xa, m*4); hipMalloc(&xb, m*4); hipMalloc(&xc, m*4); hipMalloc(&xo, m*sizeof(result));
        hipMemcpy(xa, fa.data(), m*4, hipMemcpyHostToDevice); hipMemcpy(xb, fb.data(), m*4, hipMemcpyHostToDevice); hipMemcpy(xc, fc.data(), m*4, hipMemcpyHostToDevice);
        probe32<<<1, 64>>>(xa, xb, xc, xo, m);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel32 failed\n"); return 1; }
        std::vector<result> o32(m);
        hipMemcpy(o32.data(), xo, m*sizeof(result), hipMemcpyDeviceToHost);
        printf("---- fp32 ----\n");
        for (int i = 0; i < m; i++) {
            if (!p32[i].what) continue;
            printf("%-24s %-44s -> %-24.9g flags 0x%03x %s\n", ops[p32[i].op], p32[i].what, o32[i].value, o32[i].flags, bits(o32[i].flags));
        }
    }
    return 0;
}

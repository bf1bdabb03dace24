//  How fast does one SIMD of gfx950 issue fp64 vector instructions from ONE or TWO waves, dependent or independent?
//  (round 3: is a register assignment of our own worth building for the RK4 item?  It would emit the DAG in
//  pressure order — mostly dependent chains — and count on a second wave per SIMD to fill the issue slots.)
//  Each wave runs `loops` x 256 v_fma_f64 of one shape; 256-thread blocks = one wave per SIMD of a CU.
//      hipcc --offload-arch=gfx950 -O2 fp64_issue.hip -o fp64_issue && ./fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template<int SHAPE>
__global__ void __launch_bounds__(256) chain(double *out, const int loops) {
    double a = threadIdx.x*1.0e-3, b = 1.0000001, c = 1.0e-9, d = a + 1.0, e = a + 2.0, f = a + 3.0;
    for (int l = 0; l < loops; l++) {
        if (SHAPE == 0) {           // one dependent chain
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n")
                         : "+v"(a) : "v"(b), "v"(c));
        } else if (SHAPE == 1) {    // two interleaved chains
            asm volatile(REP64("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n")
                         : "+v"(a), "+v"(d) : "v"(b), "v"(c));
        } else if (SHAPE == 2) {    // four interleaved chains
            asm volatile(REP64("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (SHAPE == 3) {    // dependent chain of v_mul_f64 / v_add_f64 alternating
            asm volatile(REP64("v_mul_f64 %0, %0, %1\n v_add_f64 %0, %0, %2\n v_mul_f64 %0, %0, %1\n v_add_f64 %0, %0, %2\n")
                         : "+v"(a) : "v"(b), "v"(c));
        } else {                    // dependent chain with an LDS round trip every 64 instructions is not measured here
        }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a + d + e + f;
}

template<int SHAPE>
static void run(const char *name, double *out) {
    const int loops = 2000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int blocks = 256*waves;
        hipEvent_t start, stop;
        hipEventCreate(&start);
        hipEventCreate(&stop);
        chain<SHAPE><<<blocks, 256>>>(out, loops);
        hipDeviceSynchronize();
        hipEventRecord(start);
        chain<SHAPE><<<blocks, 256>>>(out, loops);
        hipEventRecord(stop);
        hipEventSynchronize(stop);
        float ms = 0;
        hipEventElapsedTime(&ms, start, stop);
        const double per_wave = 256.0*loops;
        printf("%-28s %d wave(s)/SIMD: %8.3f ms  -> %.2f ns per instruction of a wave, %.2f ns per instruction issued by the SIMD\n",
               name, waves, ms, 1.0e6*ms/per_wave, 1.0e6*ms/(per_wave*waves));
    }
}

int main() {
    double *out;
    hipMalloc(&out, 256*4*256*sizeof(double));
    run<0>("one dependent fma chain", out);
    run<1>("two interleaved chains", out);
    run<2>("four interleaved chains", out);
    run<3>("dependent mul/add chain", out);
    hipFree(out);
    return 0;
}

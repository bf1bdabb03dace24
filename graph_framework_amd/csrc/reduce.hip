//------------------------------------------------------------------------------
///  @file reduce.hip
///  @brief Device-wide max reduction for converge items (hand-written, gfx950).
///
///  Replaces the reference's generated `max_reduction` kernel
///  (cuda_context.hpp:954-995: one block of 1024 threads, 32-wide shuffles, result
///  read through managed memory) and cpu_context's std::max_element
///  (cpu_context.hpp:306-322).  Here: a grid-stride pass with 16 B/lane loads,
///  a 64-lane wavefront reduction with __shfl_down (no LDS traffic), one LDS
///  word per wave, and ONE device-scope atomicMax per workgroup on an
///  order-preserving integer image of the value.  max is exact and order
///  independent, so the result is bit-identical to a serial scan.
///  HBM-bound: 8 (4) bytes per element, one pass.
//------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "converge_state.hpp"

namespace gfhip {

//  Monotone map real -> unsigned so that integer max == floating max.
__device__ __forceinline__ unsigned long long ordered(const double x) {
    const unsigned long long b = static_cast<unsigned long long> (__double_as_longlong(x));
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long ordered(const float x) {
    const unsigned int b = __float_as_uint(x);
    const unsigned int k = (b >> 31) ? ~b : (b | 0x80000000u);
    return static_cast<unsigned long long> (k);
}

template<typename T> struct alignas(2*sizeof(T)) vec2 { T a, b; };

template<typename T>
__global__ void __launch_bounds__(256)
max_reduce_kernel(const T *__restrict__ base, const unsigned long long total, const unsigned int head,
                  unsigned long long *__restrict__ result) {
    const T lowest = -__builtin_huge_val();
    T m = lowest;

//  `head` leading elements bring the pointer to 16 B alignment (caller-owned
//  buffers may be shards that start at an odd element).
    const T *in = base + head;
    const unsigned long long n = total - head;
    if (blockIdx.x == 0 && threadIdx.x < head) {
//  The same compare as everywhere below: a NaN among the head elements is not selected
//  (element 0 is handled by first_is_nan).
        const T v = base[threadIdx.x];
        m = v > m ? v : m;
    }
//  std::max_element (cpu_context.hpp:306-322) returns element 0 if that is a NaN: every later
//  `max < x` is false.  Elsewhere a NaN is never selected (`v > m` is false).
    const bool first_is_nan = total > 0 && base[0] != base[0];

//  Pairs of elements per lane per load (16 B for double).
    const unsigned long long pairs = n/2;
    const vec2<T> *in2 = reinterpret_cast<const vec2<T> *> (in);
    for (unsigned long long i = blockIdx.x*static_cast<unsigned long long> (blockDim.x) + threadIdx.x;
         i < pairs; i += gridDim.x*static_cast<unsigned long long> (blockDim.x)) {
        const vec2<T> v = in2[i];
        m = v.a > m ? v.a : m;
        m = v.b > m ? v.b : m;
    }
    if ((n & 1ull) && blockIdx.x == 0 && threadIdx.x == 0) {
        const T v = in[n - 1];
        m = v > m ? v : m;
    }

//  64-lane wavefront reduction.
    for (int offset = 32; offset > 0; offset >>= 1) {
        const T other = __shfl_down(m, offset, 64);
        m = other > m ? other : m;
    }

    __shared__ T wave_max[4];
    const unsigned int lane = threadIdx.x & 63u;
    const unsigned int wave = threadIdx.x >> 6;
    if (lane == 0) {
        wave_max[wave] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        T block = wave_max[0];
        for (unsigned int w = 1; w < (blockDim.x >> 6); w++) {
            block = wave_max[w] > block ? wave_max[w] : block;
        }
        atomicMax(result, ordered(block));
        if (blockIdx.x == 0 && first_is_nan) atomicMax(result, ~0ull >> (sizeof(T) == 8 ? 0 : 32));
    }
}

//  Device side of workflow::converge_item::run (workflow.hpp:179-205): one thread applies the
//  loop's test to the max the pass has just reduced, in the item's own precision, and either
//  stops the loop (`done`; later passes of the batch return at once) or advances its state.
//      while (|max| > tol && |last - max| > tol && |off_last - max| > tol && iterations++ < limit) {
//          last = max;  if (!(iterations%2)) off_last = max;  max = run_max(); }

template<typename T>
__global__ void converge_decide_kernel(unsigned long long *__restrict__ reduced, converge_state *__restrict__ state) {
    if (state->done) return;
    const unsigned long long key = *reduced;
    *reduced = 0ull;                                   // ready for the next pass's atomicMax
    T value;
    if (sizeof(T) == 8) {
        const unsigned long long bits = (key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key;
        value = static_cast<T> (__longlong_as_double(static_cast<long long> (bits)));
    } else {
        const unsigned int k32 = static_cast<unsigned int> (key);
        const unsigned int bits = (k32 >> 31) ? (k32 & 0x7FFFFFFFu) : ~k32;
        value = static_cast<T> (__uint_as_float(bits));
    }
    const T tolerance = static_cast<T> (state->tolerance);
    const T last = static_cast<T> (state->last), off_last = static_cast<T> (state->off_last);
    state->max_residual = static_cast<double> (value);
    state->passes++;
    const T zero = static_cast<T> (0);
    const T a = value < zero ? -value : value, t = tolerance < zero ? -tolerance : tolerance;
    const T dl = last - value, dol = off_last - value;
    bool go = a > t && (dl < zero ? -dl : dl) > t && (dol < zero ? -dol : dol) > t;
    if (go) {
        go = state->iterations < state->limit;
        state->iterations++;
    }
    if (go) {
        state->last = static_cast<double> (value);
        if (!(state->iterations%2ull)) state->off_last = static_cast<double> (value);
    } else {
        state->done = 1u;
    }
}

//  The same test applied to the maxima of a batch of passes, in order (`<name>_batch`, codegen.hpp): the loop
//  may end on any of them.  If it ends before the last pass of the batch, `extra` passes ran that the host loop
//  would not have run; the host restores the state of the beginning of the batch and redoes `batch_passes`.
template<typename T>
__global__ void converge_decide_batch_kernel(unsigned long long *__restrict__ reduced, converge_state *__restrict__ state,
                                             const unsigned int count) {
    if (state->done) return;
    const T tolerance = static_cast<T> (state->tolerance);
    const T zero = static_cast<T> (0);
    const T t = tolerance < zero ? -tolerance : tolerance;
    for (unsigned int pass = 0; pass < count; pass++) {
        const unsigned long long key = reduced[pass];
        reduced[pass] = 0ull;
        if (state->done) continue;                   // still clear the slots of the passes beyond the end
        T value;
        if (sizeof(T) == 8) {
            const unsigned long long bits = (key >> 63) ? (key & 0x7FFFFFFFFFFFFFFFull) : ~key;
            value = static_cast<T> (__longlong_as_double(static_cast<long long> (bits)));
        } else {
            const unsigned int k32 = static_cast<unsigned int> (key);
            const unsigned int bits = (k32 >> 31) ? (k32 & 0x7FFFFFFFu) : ~k32;
            value = static_cast<T> (__uint_as_float(bits));
        }
        const T last = static_cast<T> (state->last), off_last = static_cast<T> (state->off_last);
        state->max_residual = static_cast<double> (value);
        state->passes++;
        const T a = value < zero ? -value : value;
        const T dl = last - value, dol = off_last - value;
        bool go = a > t && (dl < zero ? -dl : dl) > t && (dol < zero ? -dol : dol) > t;
        if (go) {
            go = state->iterations < state->limit;
            state->iterations++;
        }
        if (go) {
            state->last = static_cast<double> (value);
            if (!(state->iterations%2ull)) state->off_last = static_cast<double> (value);
        } else {
            state->done = 1u;
            state->batch_passes = pass + 1u;
            state->extra = count - (pass + 1u);
        }
    }
}

void launch_converge_decide_batch(const bool f64, unsigned long long *reduced, void *state, const unsigned int count, hipStream_t stream) {
    if (f64) {
        hipLaunchKernelGGL(converge_decide_batch_kernel<double>, dim3(1), dim3(1), 0, stream, reduced, static_cast<converge_state *> (state), count);
    } else {
        hipLaunchKernelGGL(converge_decide_batch_kernel<float>, dim3(1), dim3(1), 0, stream, reduced, static_cast<converge_state *> (state), count);
    }
}

void launch_converge_decide(const bool f64, unsigned long long *reduced, void *state, hipStream_t stream) {
    if (f64) {
        hipLaunchKernelGGL(converge_decide_kernel<double>, dim3(1), dim3(1), 0, stream, reduced, static_cast<converge_state *> (state));
    } else {
        hipLaunchKernelGGL(converge_decide_kernel<float>, dim3(1), dim3(1), 0, stream, reduced, static_cast<converge_state *> (state));
    }
}

//  Complex outputs: the element of largest modulus, the first of equals — std::max_element with
//  std::abs as cpu_context.hpp:314-318 calls it (one workgroup, as cuda_context.hpp:954-995).
template<typename T> struct complex_pair { T re, im; };

template<typename T>
__global__ void __launch_bounds__(1024)
max_modulus_kernel(const complex_pair<T> *__restrict__ in, const unsigned long long n, complex_pair<T> *__restrict__ result) {
    T best = -1;
    unsigned long long where = n;
    for (unsigned long long i = threadIdx.x; i < n; i += blockDim.x) {
        const T modulus = sizeof(T) == 8 ? static_cast<T> (hypot(static_cast<double> (in[i].re), static_cast<double> (in[i].im)))
                                         : static_cast<T> (hypotf(static_cast<float> (in[i].re), static_cast<float> (in[i].im)));
        if (modulus > best) {                          // strictly greater: the first of equals stays
            best = modulus;
            where = i;
        }
    }
    __shared__ T moduli[1024];
    __shared__ unsigned long long places[1024];
    moduli[threadIdx.x] = best;
    places[threadIdx.x] = where;
    __syncthreads();
    for (unsigned int half = 512; half > 0; half >>= 1) {
        if (threadIdx.x < half) {
            const T other = moduli[threadIdx.x + half];
            const unsigned long long place = places[threadIdx.x + half];
            if (other > moduli[threadIdx.x] || (other == moduli[threadIdx.x] && place < places[threadIdx.x])) {
                moduli[threadIdx.x] = other;
                places[threadIdx.x] = place;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
//  std::max_element keeps element 0 when every comparison is false (all moduli NaN).
        *result = in[places[0] < n ? places[0] : 0];
    }
}

void launch_max_modulus(const void *in, const size_t n, const bool f64, void *result, hipStream_t stream) {
    if (f64) {
        hipLaunchKernelGGL(max_modulus_kernel<double>, dim3(1), dim3(1024), 0, stream,
                           static_cast<const complex_pair<double> *> (in), static_cast<unsigned long long> (n),
                           static_cast<complex_pair<double> *> (result));
    } else {
        hipLaunchKernelGGL(max_modulus_kernel<float>, dim3(1), dim3(1024), 0, stream,
                           static_cast<const complex_pair<float> *> (in), static_cast<unsigned long long> (n),
                           static_cast<complex_pair<float> *> (result));
    }
}

void launch_max_reduce(const void *in, const size_t n, const bool f64,
                       unsigned long long *result, const unsigned int num_cus, hipStream_t stream) {
    const unsigned int block = 256;
    const uintptr_t address = reinterpret_cast<uintptr_t> (in);
    const size_t element = f64 ? 8 : 4;
    size_t head = ((16 - address%16)%16)/element;
    if (head > n) head = n;
    unsigned long long want = (n/2 + block - 1)/block;
    if (want < 1) want = 1;
//  One atomicMax per workgroup lands on one address (~11 ns each, serialised): keep the grid
//  at one workgroup per CU and let the lanes stride (24 us -> 4 us for 1e6 doubles).
    const unsigned long long cap = static_cast<unsigned long long> (num_cus);
    const unsigned int grid = static_cast<unsigned int> (want < cap ? want : cap);
    if (f64) {
        hipLaunchKernelGGL(max_reduce_kernel<double>, dim3(grid), dim3(block), 0, stream,
                           static_cast<const double *> (in), static_cast<unsigned long long> (n),
                           static_cast<unsigned int> (head), result);
    } else {
        hipLaunchKernelGGL(max_reduce_kernel<float>, dim3(grid), dim3(block), 0, stream,
                           static_cast<const float *> (in), static_cast<unsigned long long> (n),
                           static_cast<unsigned int> (head), result);
    }
}

}  // namespace gfhip

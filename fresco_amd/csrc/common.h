// Shared helpers for the gfx950 kernels of libfresco_hip.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fresco_hip.h"

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace fresco {

void set_last_error(hipError_t e);

// Every launch wrapper ends with this: maps a failed launch to FRESCO_ELAUNCH.
static inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error(e);
        return FRESCO_ELAUNCH;
    }
    return FRESCO_OK;
}

// Opt-in kernel timing (fresco_prof_*): brackets selected launches with HIP events on the launch stream.
struct ProfScope {
    bool on;
    hipStream_t st;
    ProfScope(int tag, int a, int b, int c, int d, hipStream_t s);
    ~ProfScope();
};

// true while fresco_prof_enable() is in effect (kernels that would otherwise overlap on two streams then run on one)
bool prof_active();

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

template <typename T>
static inline T* carve(char*& p, size_t count) {
    T* r = reinterpret_cast<T*>(p);
    p += align_up(count * sizeof(T), 256);
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace fresco

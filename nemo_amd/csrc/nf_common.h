// Shared definitions for the nemofilt HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>

#define NF_OK 0
#define NF_ERR_ARG (-1)
#define NF_ERR_HIP (-2)
#define NF_ERR_SIZE (-3)
#define NF_ERR_NOMEM (-4)

#define NF_F32 0
#define NF_F64 1

void nf_set_error(const std::string& msg);
int nf_hip_fail(hipError_t e, const char* what);

#define NF_HIP(call)                                            \
    do {                                                        \
        hipError_t _e = (call);                                 \
        if (_e != hipSuccess) return nf_hip_fail(_e, #call);    \
    } while (0)

#define NF_CHECK_LAUNCH(name)                                   \
    do {                                                        \
        hipError_t _e = hipGetLastError();                      \
        if (_e != hipSuccess) return nf_hip_fail(_e, name);     \
    } while (0)

template <typename T>
struct cpx {
    T x, y;
};
template <>
struct __attribute__((aligned(8))) cpx<float> {
    float x, y;
};
template <>
struct __attribute__((aligned(16))) cpx<double> {
    double x, y;
};

template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b) {
    cpx<T> r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
template <typename T>
__device__ __forceinline__ cpx<T> cadd(cpx<T> a, cpx<T> b) {
    cpx<T> r;
    r.x = a.x + b.x;
    r.y = a.y + b.y;
    return r;
}
template <typename T>
__device__ __forceinline__ cpx<T> csub(cpx<T> a, cpx<T> b) {
    cpx<T> r;
    r.x = a.x - b.x;
    r.y = a.y - b.y;
    return r;
}
template <typename T>
__device__ __forceinline__ cpx<T> cconj(cpx<T> a) {
    cpx<T> r;
    r.x = a.x;
    r.y = -a.y;
    return r;
}
// multiply by -i (forward DFT quarter turn)
template <typename T>
__device__ __forceinline__ cpx<T> cmulnegi(cpx<T> a) {
    cpx<T> r;
    r.x = a.y;
    r.y = -a.x;
    return r;
}

static inline int nf_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

static const int kMaxLds = 160 * 1024;      // LDS of a gfx950 CU: the most a line-transform workgroup may ask for

// float64 <-> unsigned 64-bit key with the same order (integer atomicMax / atomicMin on doubles)
__device__ __forceinline__ unsigned long long nf_ordered_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double nf_key_value(unsigned long long k) {
    const unsigned long long b = (k & 0x8000000000000000ULL) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
    return __longlong_as_double((long long)b);
}
#define NF_SN_CHUNK 1024   // pixels per chunk of the S/N range table (nf_finalize) = the labelling passes' chunk

// ---- optional per-entry-point timing with HIP events (nf_profile_* in include/nemofilt.h) ----
namespace nf {
enum ProfTag {
    PT_FFT_ROWS_F64 = 0,
    PT_FFT_COLS_F64,
    PT_FFT_ROWS_F32,
    PT_FFT_COLS_F32,
    PT_CROSS_POWER,
    PT_GAUSSIAN,
    PT_BUILD_FILTER,
    PT_APPLY_FILTER,
    PT_WINDOW_MUL,
    PT_PAINT,
    PT_SCALE_MASK,
    PT_NOISE_MAP,
    PT_MEDIAN_STACK,
    PT_EDGE_CHECK,
    PT_FINALIZE,
    PT_RESTORE_ZEROS,
    PT_SUBTRACT,
    PT_CONVOLVE2D,
    PT_LABEL,
    PT_OBJECT_STATS,
    PT_MAP_MAX,
    PT_RING_FLAGS,
    PT_SPLINE,
    PT_AXIS_FILTER_ROWS,
    PT_AXIS_FILTER_COLS,
    PT_COUNT
};
// Records an event pair around the work enqueued on `s` during its lifetime when profiling is on (no-op otherwise).
// `bytes` = algorithmic HBM bytes of that work (inputs read once + outputs written once).
struct ProfScope {
    ProfScope(int tag, hipStream_t s, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};
}  // namespace nf

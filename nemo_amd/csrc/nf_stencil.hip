// Neighbourhood kernels with scipy.ndimage semantics ('reflect' borders = edge sample repeated).
//
//  k_gauss_axis   K5/K14  ndimage.gaussian_filter  (nemo/filters.py:580, nemo/maps.py:1380): separable,
//                         axis 0 then axis 1, float64 accumulation in scipy's symmetric-kernel order
//                         (centre tap, then pairs from the outermost inwards), intermediate stored in the
//                         map dtype between the two passes.
//  k_convolve2d   K13     ndimage.convolve (nemo/filters.py:1193): odd kernel, float64 accumulation over the
//                         flipped kernel in raster order, taps with |w| <= DBL_EPSILON skipped, output in the map dtype.
//  k_edge_*       K12     ndimage.rank_filter(abs(m + (1 - psMask)), 0, size=(t,t)) > 0 (nemo/filters.py:735,
//                         1139): a moving-window minimum > 0 is an AND of "pixel != 0" over the window
//                         [i - t//2, i + t - t//2 - 1]; separable, exact.
#include "nf_common.h"
#include "nf_kernels.h"

#include <cfloat>

namespace nf {

__device__ __forceinline__ int reflect_index(int i, int n) {
    // scipy 'reflect' (numpy 'symmetric'): ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...
    if (n == 1) return 0;
    int period = 2 * n;
    i = i % period;
    if (i < 0) i += period;
    return (i < n) ? i : period - 1 - i;
}

template <typename T, int AXIS>
__global__ void k_gauss_axis(const T* __restrict__ in, T* __restrict__ out, int H, int W, const double* __restrict__ w, int rad) {
    const long long plane = (long long)H * W;
    const T* src = in + (long long)blockIdx.y * plane;
    T* dst = out + (long long)blockIdx.y * plane;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < plane; p += (long long)gridDim.x * blockDim.x) {
        int y = (int)(p / W), x = (int)(p % W);
        double acc = __dmul_rn((double)src[p], w[rad]);
        if (AXIS == 0) {
            bool interior = (y - rad >= 0) && (y + rad < H);
            for (int j = -rad; j < 0; ++j) {
                int ya = interior ? y + j : reflect_index(y + j, H);
                int yb = interior ? y - j : reflect_index(y - j, H);
                double pair = __dadd_rn((double)src[(long long)ya * W + x], (double)src[(long long)yb * W + x]);
                acc = __dadd_rn(acc, __dmul_rn(pair, w[rad + j]));
            }
        } else {
            bool interior = (x - rad >= 0) && (x + rad < W);
            const T* row = src + (long long)y * W;
            for (int j = -rad; j < 0; ++j) {
                int xa = interior ? x + j : reflect_index(x + j, W);
                int xb = interior ? x - j : reflect_index(x - j, W);
                double pair = __dadd_rn((double)row[xa], (double)row[xb]);
                acc = __dadd_rn(acc, __dmul_rn(pair, w[rad + j]));
            }
        }
        dst[p] = (T)acc;
    }
}

static inline int grid_for(long long n, int threads) {
    long long g = (n + threads - 1) / threads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// The same two passes with each input value fetched once instead of 2 rad + 1 times.  Same arithmetic, same order of the
// additions (centre tap, then the pairs from the outermost inwards): bit for bit what k_gauss_axis gives.
//  along y (axis 0): a thread owns one column of a strip of NF_GA_STRIP rows and holds its NF_GA_STRIP + 2 RAD inputs in registers
//                    (lanes = consecutive columns: every load is a coalesced row segment);
//  along x (axis 1): a workgroup stages 1024 + 2 rad values of one row in LDS, every thread forms four outputs from there.
#define NF_GA_STRIP 16
template <typename T, int RAD>
__global__ void __launch_bounds__(256) k_gauss_cols_strip(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                          const double* __restrict__ w) {
    const long long plane = (long long)H * W;
    const T* src = in + (long long)blockIdx.z * plane;
    T* dst = out + (long long)blockIdx.z * plane;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const int y0 = blockIdx.y * NF_GA_STRIP;
    const bool interior = (y0 - RAD >= 0) && (y0 + NF_GA_STRIP - 1 + RAD < H);
    double v[NF_GA_STRIP + 2 * RAD];
#pragma unroll
    for (int i = 0; i < NF_GA_STRIP + 2 * RAD; ++i) {
        int yy = y0 - RAD + i;
        if (!interior) yy = reflect_index(yy, H);
        v[i] = (double)src[(long long)yy * W + x];
    }
    double wr[RAD + 1];
#pragma unroll
    for (int j = 0; j <= RAD; ++j) wr[j] = w[j];
#pragma unroll
    for (int o = 0; o < NF_GA_STRIP; ++o) {
        if (y0 + o >= H) break;
        double acc = __dmul_rn(v[o + RAD], wr[RAD]);
#pragma unroll
        for (int j = -RAD; j < 0; ++j) {
            const double pair = __dadd_rn(v[o + RAD + j], v[o + RAD - j]);
            acc = __dadd_rn(acc, __dmul_rn(pair, wr[RAD + j]));
        }
        dst[(long long)(y0 + o) * W + x] = (T)acc;
    }
}

#define NF_GA_SEG 1024
template <typename T>
__global__ void __launch_bounds__(256) k_gauss_rows_lds(const T* __restrict__ in, T* __restrict__ out, int H, int W,
                                                        const double* __restrict__ w, int rad) {
    extern __shared__ __align__(16) unsigned char nf_gsmem[];
    T* tile = (T*)nf_gsmem;
    const long long plane = (long long)H * W;
    const int y = blockIdx.y, x0 = blockIdx.x * NF_GA_SEG;
    const T* row = in + (long long)blockIdx.z * plane + (long long)y * W;
    T* drow = out + (long long)blockIdx.z * plane + (long long)y * W;
    const int span = NF_GA_SEG + 2 * rad;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int xx = x0 - rad + i;
        tile[i] = row[(xx >= 0 && xx < W) ? xx : reflect_index(xx, W)];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NF_GA_SEG / 256; ++k) {
        const int c = k * 256 + (int)threadIdx.x;
        if (x0 + c >= W) break;
        const T* t = tile + c + rad;
        double acc = __dmul_rn((double)t[0], w[rad]);
        for (int j = -rad; j < 0; ++j) {
            const double pair = __dadd_rn((double)t[j], (double)t[-j]);
            acc = __dadd_rn(acc, __dmul_rn(pair, w[rad + j]));
        }
        drow[x0 + c] = (T)acc;
    }
}

template <typename T>
static int gaussian_filter_t(const T* in, T* tmp, T* out, int H, int W, int batch, const double* wY, int radY, const double* wX,
                             int radX, hipStream_t s) {
    dim3 b(256);
    // along y: the register strip for the radius of the noise-power smoothing (sigma 3 -> 12 taps each side)
    if (radY == 12 && H <= 65535 * NF_GA_STRIP && batch <= 65535) {
        dim3 g(nf_div_up(W, 256), nf_div_up(H, NF_GA_STRIP), batch);
        hipLaunchKernelGGL((k_gauss_cols_strip<T, 12>), g, b, 0, s, in, tmp, H, W, wY);
    } else {
        dim3 g(grid_for((long long)H * W, 256), batch);
        hipLaunchKernelGGL((k_gauss_axis<T, 0>), g, b, 0, s, in, tmp, H, W, wY, radY);
    }
    if (radX <= 512 && H <= 65535 && batch <= 65535) {
        dim3 g(nf_div_up(W, NF_GA_SEG), H, batch);
        hipLaunchKernelGGL((k_gauss_rows_lds<T>), g, b, (size_t)(NF_GA_SEG + 2 * radX) * sizeof(T), s, tmp, out, H, W, wX, radX);
    } else {
        dim3 g(grid_for((long long)H * W, 256), batch);
        hipLaunchKernelGGL((k_gauss_axis<T, 1>), g, b, 0, s, tmp, out, H, W, wX, radX);
    }
    NF_CHECK_LAUNCH("k_gauss_axis");
    return NF_OK;
}

int gaussian_filter(int rtype, const void* in, void* tmp, void* out, int H, int W, int batch, const double* wY, int radY,
                    const double* wX, int radX, hipStream_t s) {
    if (rtype == NF_F64)
        return gaussian_filter_t<double>((const double*)in, (double*)tmp, (double*)out, H, W, batch, wY, radY, wX, radX, s);
    return gaussian_filter_t<float>((const float*)in, (float*)tmp, (float*)out, H, W, batch, wY, radY, wX, radX, s);
}

template <typename T>
__global__ void k_subtract(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        out[p] = a[p] - b[p];
}

int subtract(int rtype, const void* a, const void* b, void* out, long long n, hipStream_t s) {
    dim3 g(grid_for(n, 256)), blk(256);
    if (rtype == NF_F64)
        hipLaunchKernelGGL(k_subtract<double>, g, blk, 0, s, (const double*)a, (const double*)b, (double*)out, n);
    else
        hipLaunchKernelGGL(k_subtract<float>, g, blk, 0, s, (const float*)a, (const float*)b, (float*)out, n);
    NF_CHECK_LAUNCH("k_subtract");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
#define NF_CONV_TILE 32
template <typename T>
__global__ void k_convolve2d(const T* __restrict__ in, T* __restrict__ out, int H, int W, const double* __restrict__ kf,
                             int kh, int kw) {
    // LDS tile of the input (map dtype) with halo; one output pixel per thread
    extern __shared__ __align__(16) unsigned char nf_smem[];
    T* tile = (T*)nf_smem;
    const int hy = kh / 2, hx = kw / 2;
    const int tw = NF_CONV_TILE + kw - 1, th = NF_CONV_TILE + kh - 1;
    const int x0 = blockIdx.x * NF_CONV_TILE, y0 = blockIdx.y * NF_CONV_TILE;
    const int tid = threadIdx.y * NF_CONV_TILE + threadIdx.x;
    for (int i = tid; i < tw * th; i += NF_CONV_TILE * NF_CONV_TILE) {
        int ty = i / tw, tx = i % tw;
        int gy = reflect_index(y0 + ty - hy, H), gx = reflect_index(x0 + tx - hx, W);
        tile[i] = in[(long long)gy * W + gx];
    }
    __syncthreads();
    const int x = x0 + threadIdx.x, y = y0 + threadIdx.y;
    if (x >= W || y >= H) return;
    double acc = 0.0;
    for (int fy = 0; fy < kh; ++fy) {
        const T* trow = tile + (threadIdx.y + fy) * tw + threadIdx.x;
        const double* krow = kf + fy * kw;
        for (int fx = 0; fx < kw; ++fx) {
            double wv = krow[fx];
            if (fabs(wv) > DBL_EPSILON) acc = __dadd_rn(acc, __dmul_rn((double)trow[fx], wv));
        }
    }
    out[(long long)y * W + x] = (T)acc;
}

int convolve2d(int rtype, const void* in, void* out, int H, int W, const double* kernFlipped, int kh, int kw, hipStream_t s) {
    if ((kh % 2) == 0 || (kw % 2) == 0) {
        nf_set_error("convolve2d: kernel dimensions must be odd");
        return NF_ERR_ARG;
    }
    size_t lds = (size_t)(NF_CONV_TILE + kw - 1) * (NF_CONV_TILE + kh - 1) * (rtype == NF_F64 ? sizeof(double) : sizeof(float));
    if (lds > 64 * 1024) {
        nf_set_error("convolve2d: kernel too large for the LDS tile");
        return NF_ERR_SIZE;
    }
    dim3 g(nf_div_up(W, NF_CONV_TILE), nf_div_up(H, NF_CONV_TILE)), b(NF_CONV_TILE, NF_CONV_TILE);
    if (rtype == NF_F64)
        hipLaunchKernelGGL(k_convolve2d<double>, g, b, lds, s, (const double*)in, (double*)out, H, W, kernFlipped, kh, kw);
    else
        hipLaunchKernelGGL(k_convolve2d<float>, g, b, lds, s, (const float*)in, (float*)out, H, W, kernFlipped, kh, kw);
    NF_CHECK_LAUNCH("k_convolve2d");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
template <typename VT>
__global__ void k_edge_rows(const VT* __restrict__ f, const unsigned char* __restrict__ ps, unsigned char* __restrict__ out, int H, int W,
                            int size) {
    const long long n = (long long)H * W;
    const int lo = size / 2, hi = size - size / 2 - 1;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y = (int)(p / W), x = (int)(p % W);
        unsigned char ok = 1;
        for (int j = -lo; j <= hi && ok; ++j) {
            int xx = reflect_index(x + j, W);
            long long q = (long long)y * W + xx;
            VT v = f[q] + (VT)(unsigned char)(1 - (ps ? ps[q] : 1));
            if (!(fabs((double)v) > 0.0)) ok = 0;
        }
        out[p] = ok;
    }
}

__global__ void k_edge_cols(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int H, int W, int size) {
    const long long n = (long long)H * W;
    const int lo = size / 2, hi = size - size / 2 - 1;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        int y = (int)(p / W), x = (int)(p % W);
        unsigned char ok = 1;
        for (int j = -lo; j <= hi && ok; ++j) {
            int yy = reflect_index(y + j, H);
            if (!in[(long long)yy * W + x]) ok = 0;
        }
        out[p] = ok;
    }
}

int edge_check(int vtype, const void* filtered, const unsigned char* psMask, unsigned char* tmp, unsigned char* out, int H, int W,
               int size, hipStream_t s) {
    dim3 g(grid_for((long long)H * W, 256)), b(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_edge_rows<double>, g, b, 0, s, (const double*)filtered, psMask, tmp, H, W, size);
    else
        hipLaunchKernelGGL(k_edge_rows<float>, g, b, 0, s, (const float*)filtered, psMask, tmp, H, W, size);
    hipLaunchKernelGGL(k_edge_cols, g, b, 0, s, tmp, out, H, W, size);
    NF_CHECK_LAUNCH("k_edge");
    return NF_OK;
}

}  // namespace nf

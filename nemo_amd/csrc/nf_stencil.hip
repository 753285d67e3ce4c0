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

int gaussian_filter(int rtype, const void* in, void* tmp, void* out, int H, int W, int batch, const double* wY, int radY,
                    const double* wX, int radX, hipStream_t s) {
    dim3 g(grid_for((long long)H * W, 256), batch), b(256);
    if (rtype == NF_F64) {
        hipLaunchKernelGGL((k_gauss_axis<double, 0>), g, b, 0, s, (const double*)in, (double*)tmp, H, W, wY, radY);
        hipLaunchKernelGGL((k_gauss_axis<double, 1>), g, b, 0, s, (const double*)tmp, (double*)out, H, W, wX, radX);
    } else {
        hipLaunchKernelGGL((k_gauss_axis<float, 0>), g, b, 0, s, (const float*)in, (float*)tmp, H, W, wY, radY);
        hipLaunchKernelGGL((k_gauss_axis<float, 1>), g, b, 0, s, (const float*)tmp, (float*)out, H, W, wX, radX);
    }
    NF_CHECK_LAUNCH("k_gauss_axis");
    return NF_OK;
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

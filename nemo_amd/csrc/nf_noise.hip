// Local-RMS noise map (K10): MapFilter.makeNoiseMap, grid mode (nemo/filters.py:407-478).
//
// Two kernels:
//  k_noise_cells  one workgroup per grid cell (cell widened by `overlap` pixels on each side): counts the
//                 valid (non-zero) pixels, finds the weight-percentile bin edges of the cell (radix select on
//                 the float64 weight keys, all ranks in one sweep per digit), then for every bin the 3-sigma
//                 clipped standard deviation (10 iterations, test |v| < |mean + 3 rms|, population std,
//                 two-pass mean / variance with wavefront + LDS tree reductions in float64).
//  k_noise_paint  one thread per pixel: the reference writes cells in (i, k, bin) loop order and later cells
//                 overwrite earlier ones (nemo/filters.py:478), so each pixel takes the value of the LAST cell
//                 in that order whose window contains it and whose bin for the pixel's weight has rms > 0.
//
// The cell windows stay L2-resident between the passes of k_noise_cells (<= 160x160 pixels for 40' cells).
#include "nf_common.h"
#include "nf_kernels.h"

namespace nf {

#define NF_NOISE_THREADS 256
#define NF_MAXBINS 16

__device__ __forceinline__ double block_sum(double v, double* red) {
    // wavefront shuffle reduction, then LDS across the 4 waves; deterministic for a fixed block size
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < NF_NOISE_THREADS / 64; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ double block_min(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < NF_NOISE_THREADS / 64; ++i) t = fmin(t, red[i]);
    return t;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < NF_NOISE_THREADS / 64; ++i) t = fmax(t, red[i]);
    return t;
}

__device__ __forceinline__ unsigned long long key_of(double w) {
    unsigned long long b = (unsigned long long)__double_as_longlong(w);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double unkey(unsigned long long k) {
    unsigned long long b = (k & 0x8000000000000000ULL) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
    return __longlong_as_double((long long)b);
}

struct CellWin {
    int y0, y1, x0, x1;
};

__device__ __forceinline__ CellWin cell_window(const NoiseGrid& g, int cell, int H, int W) {
    int i = cell / g.nx, k = cell % g.nx;
    CellWin c;
    c.y0 = max(g.yEdges[i] - g.overlap, 0);
    c.y1 = min(g.yEdges[i + 1] + g.overlap, H);
    c.x0 = max(g.xEdges[k] - g.overlap, 0);
    c.x1 = min(g.xEdges[k + 1] + g.overlap, W);
    return c;
}

// per-cell record: [0] active, [1 .. nb+1] bin edges, [nb+2 .. 2nb+1] rms per bin
__host__ __device__ inline int cell_stride(int nb) { return 2 * nb + 2; }

template <typename VT, typename WT>
__global__ void k_noise_cells(const VT* __restrict__ v, const WT* __restrict__ wts, int H, int W, NoiseGrid g,
                              double* __restrict__ stats) {
    const double* quantiles = g.quantiles;
    __shared__ double red[NF_NOISE_THREADS / 64];
    __shared__ unsigned int hist[2 * NF_MAXBINS][256];
    __shared__ unsigned long long prefix[2 * NF_MAXBINS];
    __shared__ unsigned int rankLeft[2 * NF_MAXBINS];
    const int cell = blockIdx.x;
    const int nb = g.numBins;
    double* rec = stats + (long long)cell * cell_stride(nb);
    const CellWin c = cell_window(g, cell, H, W);
    const int ww = c.x1 - c.x0, wh = c.y1 - c.y0, npx = ww * wh;
    const int tid = threadIdx.x;

    // ---- valid pixels, weight range ----
    double cnt = 0.0, wmin = INFINITY, wmax = -INFINITY;
    for (int i = tid; i < npx; i += NF_NOISE_THREADS) {
        long long q = (long long)(c.y0 + i / ww) * W + c.x0 + i % ww;
        if (v[q] != (VT)0) {
            cnt += 1.0;
            double wv = (double)wts[q];
            wmin = fmin(wmin, wv);
            wmax = fmax(wmax, wv);
        }
    }
    const double ngood = block_sum(cnt, red);
    if (ngood == 0.0) {
        if (tid == 0) rec[0] = 0.0;
        return;
    }
    wmin = block_min(wmin, red);
    wmax = block_max(wmax, red);

    // ---- bin edges: np.percentile(weights[good], p) (linear method), last edge = max + 1e-6 ----
    if (nb == 1) {
        if (tid == 0) {
            rec[1] = wmin;  // percentile 0 = minimum
            rec[2] = wmax + 1e-6;
        }
    } else {
        // ranks needed: floor(vi) and floor(vi)+1 for each quantile, vi = n*q + (1 - q) - 1 (numpy's expression)
        const int nr = 2 * nb;
        if (tid < nr) {
            double q = quantiles[tid / 2];
            double vi = __dsub_rn(__dadd_rn(__dmul_rn(ngood, q), __dsub_rn(1.0, q)), 1.0);
            double fl = floor(vi);
            long long r = (long long)fl + (tid & 1);
            if (r > (long long)ngood - 1) r = (long long)ngood - 1;
            if (r < 0) r = 0;
            rankLeft[tid] = (unsigned int)r;
            prefix[tid] = 0ULL;
        }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int i = tid; i < nr * 256; i += NF_NOISE_THREADS) (&hist[0][0])[i] = 0u;
            __syncthreads();
            for (int i = tid; i < npx; i += NF_NOISE_THREADS) {
                long long q = (long long)(c.y0 + i / ww) * W + c.x0 + i % ww;
                if (v[q] != (VT)0) {
                    unsigned long long key = key_of((double)wts[q]);
                    unsigned long long hiBits = (shift == 56) ? 0ULL : (key >> (shift + 8));
                    unsigned int digit = (unsigned int)((key >> shift) & 0xFFULL);
                    for (int rk = 0; rk < nr; ++rk)
                        if (hiBits == prefix[rk]) atomicAdd(&hist[rk][digit], 1u);
                }
            }
            __syncthreads();
            if (tid < nr) {
                unsigned int left = rankLeft[tid], acc = 0;
                int d = 0;
                for (; d < 256; ++d) {
                    unsigned int h = hist[tid][d];
                    if (left < acc + h) break;
                    acc += h;
                }
                rankLeft[tid] = left - acc;
                prefix[tid] = (prefix[tid] << 8) | (unsigned long long)d;
            }
            __syncthreads();
        }
        if (tid < nb) {
            double q = quantiles[tid];
            double vi = __dsub_rn(__dadd_rn(__dmul_rn(ngood, q), __dsub_rn(1.0, q)), 1.0);
            double t = __dsub_rn(vi, floor(vi));
            double a = unkey(prefix[2 * tid]), b = unkey(prefix[2 * tid + 1]);
            double diff = __dsub_rn(b, a);
            double lerp = __dadd_rn(a, __dmul_rn(diff, t));
            if (t >= 0.5) lerp = __dsub_rn(b, __dmul_rn(diff, __dsub_rn(1.0, t)));
            rec[1 + tid] = lerp;
        }
        if (tid == 0) rec[1 + nb] = wmax + 1e-6;
    }
    __syncthreads();
    __threadfence_block();

    // ---- per-bin clipped standard deviation ----
    for (int b = 0; b < nb; ++b) {
        const double lo = rec[1 + b], hi = rec[2 + b];
        // first pass: all valid pixels of the bin
        double thr = INFINITY;  // |v| < thr ; the first evaluation takes everything
        double mean = 0.0, rms = 0.0;
        bool have = false;
        for (int it = 0; it <= 10; ++it) {
            double s = 0.0, n = 0.0;
            for (int i = tid; i < npx; i += NF_NOISE_THREADS) {
                long long q = (long long)(c.y0 + i / ww) * W + c.x0 + i % ww;
                double val = (double)v[q];
                double wv = (double)wts[q];
                if (val != 0.0 && wv >= lo && wv < hi && (it == 0 || fabs(val) < thr)) {
                    s += val;
                    n += 1.0;
                }
            }
            s = block_sum(s, red);
            n = block_sum(n, red);
            if (n == 0.0) {
                if (it == 0) break;   // empty bin -> rms 0
                continue;             // "if mask.sum() > 0" fails: keep previous mean / rms, same threshold next time
            }
            const double m = s / n;
            double ss = 0.0;
            for (int i = tid; i < npx; i += NF_NOISE_THREADS) {
                long long q = (long long)(c.y0 + i / ww) * W + c.x0 + i % ww;
                double val = (double)v[q];
                double wv = (double)wts[q];
                if (val != 0.0 && wv >= lo && wv < hi && (it == 0 || fabs(val) < thr)) {
                    double d = val - m;
                    ss += d * d;
                }
            }
            ss = block_sum(ss, red);
            const double r = sqrt(ss / n);
            const bool same = have && (m == mean) && (r == rms);
            mean = m;
            rms = r;
            have = true;
            thr = fabs(mean + 3.0 * rms);
            if (same) break;          // fixed point: all remaining iterations reproduce these values
        }
        if (tid == 0) rec[2 + nb + b] = have ? rms : 0.0;
    }
    if (tid == 0) rec[0] = 1.0;
}

template <typename WT>
__global__ void k_noise_paint(const WT* __restrict__ wts, float* __restrict__ rmsMap, int H, int W, NoiseGrid g,
                              const double* __restrict__ stats) {
    const long long n = (long long)H * W;
    const int nb = g.numBins;
    const int stride = cell_stride(nb);
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(p / W), x = (int)(p % W);
        const double wv = (double)wts[p];
        float outv = 0.0f;
        bool done = false;
        for (int i = g.ny - 1; i >= 0 && !done; --i) {
            if (y < g.yEdges[i] - g.overlap || y >= g.yEdges[i + 1] + g.overlap) continue;
            for (int k = g.nx - 1; k >= 0 && !done; --k) {
                if (x < g.xEdges[k] - g.overlap || x >= g.xEdges[k + 1] + g.overlap) continue;
                const double* rec = stats + (long long)(i * g.nx + k) * stride;
                if (rec[0] == 0.0) continue;
                for (int b = nb - 1; b >= 0; --b) {
                    if (wv >= rec[1 + b] && wv < rec[2 + b]) {
                        double r = rec[2 + nb + b];
                        if (r > 0.0) {
                            outv = (float)r;
                            done = true;
                        }
                        break;
                    }
                }
            }
        }
        rmsMap[p] = outv;
    }
}

// np.median over the leading axis of a [N][n] stack (N <= 8): sort in registers; even N averages the middle pair.
template <typename T>
__global__ void k_median_stack(const T* __restrict__ in, T* __restrict__ out, int N, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        T v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (i < N) ? in[(long long)i * n + p] : (T)INFINITY;
#pragma unroll
        for (int i = 1; i < 8; ++i) {
#pragma unroll
            for (int j = 7; j >= 1; --j) {
                if (j <= 8 - i) {
                    T a = v[j - 1], b = v[j];
                    if (b < a) {
                        v[j - 1] = b;
                        v[j] = a;
                    }
                }
            }
        }
        // bubble passes above leave v ascending
        T res = (T)0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (N % 2 == 1 && i == N / 2) res = v[i];
            if (N % 2 == 0 && i == N / 2) res = (v[i - (i > 0 ? 1 : 0)] + v[i]) / (T)2;   // np.mean of the middle pair
        }
        out[p] = res;
    }
}

int median_stack(int rtype, const void* in, void* out, int N, long long n, hipStream_t s) {
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (rtype == NF_F64)
        hipLaunchKernelGGL(k_median_stack<double>, dim3((int)g), dim3(256), 0, s, (const double*)in, (double*)out, N, n);
    else
        hipLaunchKernelGGL(k_median_stack<float>, dim3((int)g), dim3(256), 0, s, (const float*)in, (float*)out, N, n);
    NF_CHECK_LAUNCH("k_median_stack");
    return NF_OK;
}

size_t noise_workspace_bytes(const NoiseGrid& g) {
    return (size_t)g.ny * g.nx * cell_stride(g.numBins) * sizeof(double);
}

int noise_map(int vtype, const void* filtered, const void* weights, int wtype, float* rms, int H, int W, const NoiseGrid& g,
              void* cellStats, size_t cellStatsBytes, hipStream_t s) {
    if (g.numBins < 1 || g.numBins > NF_MAXBINS) {
        nf_set_error("noise_map: numNoiseBins must be in 1..16");
        return NF_ERR_ARG;
    }
    if (g.estimator != 0) {
        nf_set_error("noise_map: only the default clipped-std estimator is implemented on the GPU");
        return NF_ERR_ARG;
    }
    if (cellStatsBytes < noise_workspace_bytes(g)) {
        nf_set_error("noise_map: workspace too small");
        return NF_ERR_SIZE;
    }
    double* stats = (double*)cellStats;
    dim3 gc(g.ny * g.nx), bc(NF_NOISE_THREADS);
#define NF_NC(VT, WT) \
    hipLaunchKernelGGL((k_noise_cells<VT, WT>), gc, bc, 0, s, (const VT*)filtered, (const WT*)weights, H, W, g, stats)
    if (vtype == NF_F64 && wtype == NF_F64) NF_NC(double, double);
    else if (vtype == NF_F64 && wtype == NF_F32) NF_NC(double, float);
    else if (vtype == NF_F32 && wtype == NF_F64) NF_NC(float, double);
    else NF_NC(float, float);
#undef NF_NC
    NF_CHECK_LAUNCH("k_noise_cells");
    long long n = (long long)H * W;
    long long gp = (n + 255) / 256;
    if (gp > 4096) gp = 4096;
    if (wtype == NF_F64)
        hipLaunchKernelGGL(k_noise_paint<double>, dim3((int)gp), dim3(256), 0, s, (const double*)weights, rms, H, W, g, stats);
    else
        hipLaunchKernelGGL(k_noise_paint<float>, dim3((int)gp), dim3(256), 0, s, (const float*)weights, rms, H, W, g, stats);
    NF_CHECK_LAUNCH("k_noise_paint");
    return NF_OK;
}

}  // namespace nf

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
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
// Cross-lane moves on the vector ALU (DPP) instead of the LDS crossbar (ds_bpermute, which __shfl_down lowers to): a
// float64 is moved as its two halves.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
// after this every lane of a row of 16 holds the sum of its row
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_f64<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v);   // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v);   // row_mirror
    return v;
}
// lane 63 ends up with the sum over the wavefront
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v = row16_sum(v);
    v += dpp_f64<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    v += dpp_f64<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}

// three sums in one pass: one barrier pair instead of three (red3: 3 * 16 doubles; at most 16 wavefronts)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red3) {
    a = wave_sum_lane63(a);
    b = wave_sum_lane63(b);
    c = wave_sum_lane63(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 63) {
        red3[wave] = a;
        red3[16 + wave] = b;
        red3[32 + wave] = c;
    }
    __syncthreads();
    const int k = lane & 15;
    a = row16_sum(k < nw ? red3[k] : 0.0);
    b = row16_sum(k < nw ? red3[16 + k] : 0.0);
    c = row16_sum(k < nw ? red3[32 + k] : 0.0);
}
__device__ __forceinline__ double block_min(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmin(t, red[i]);
    return t;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]);
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

// Iterates over the pixels of a cell window with the lanes of a wavefront along x (coalesced) and the wavefronts
// of the workgroup striding over rows: no per-pixel division.
#define NF_CELL_LOOP(c, ww, wh, BODY)                                            \
    for (int _r = (threadIdx.x >> 6); _r < (wh); _r += NF_NOISE_THREADS / 64) {  \
        const long long _row = (long long)((c).y0 + _r) * W + (c).x0;            \
        for (int _x = (threadIdx.x & 63); _x < (ww); _x += 64) {                 \
            const long long q = _row + _x;                                       \
            BODY                                                                 \
        }                                                                        \
    }

template <typename VT, typename WT, bool ONEBIN>
__global__ void k_noise_cells(const VT* __restrict__ v, const WT* __restrict__ wts, int H, int W, NoiseGrid g,
                              double* __restrict__ stats) {
    const double* quantiles = g.quantiles;
    __shared__ double red[16];
    __shared__ unsigned int hist[ONEBIN ? 2 : 2 * NF_MAXBINS][256];
    __shared__ unsigned long long prefix[2 * NF_MAXBINS];
    __shared__ unsigned int rankLeft[2 * NF_MAXBINS];
    __shared__ double edges[NF_MAXBINS + 1];
    const int cell = blockIdx.x;
    const int nb = ONEBIN ? 1 : g.numBins;
    double* rec = stats + (long long)cell * cell_stride(nb);
    const CellWin c = cell_window(g, cell, H, W);
    const int ww = c.x1 - c.x0, wh = c.y1 - c.y0;
    const int tid = threadIdx.x;

    // ---- valid pixels, weight range ----
    double cnt = 0.0, wmin = INFINITY, wmax = -INFINITY;
    NF_CELL_LOOP(c, ww, wh, {
        if (v[q] != (VT)0) {
            cnt += 1.0;
            double wv = (double)wts[q];
            wmin = fmin(wmin, wv);
            wmax = fmax(wmax, wv);
        }
    })
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
            edges[0] = wmin;  // percentile 0 = minimum
            edges[1] = wmax + 1e-6;
        }
    } else if (!ONEBIN) {
        // ranks needed: floor(vi) and floor(vi)+1 for each quantile, vi = (n - 1) * q (numpy's expression for the 'linear' method)
        const int nr = 2 * nb;
        if (tid < nr) {
            double q = quantiles[tid / 2];
            double vi = __dmul_rn(__dsub_rn(ngood, 1.0), q);
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
            NF_CELL_LOOP(c, ww, wh, {
                if (v[q] != (VT)0) {
                    unsigned long long key = key_of((double)wts[q]);
                    unsigned long long hiBits = (shift == 56) ? 0ULL : (key >> (shift + 8));
                    unsigned int digit = (unsigned int)((key >> shift) & 0xFFULL);
                    for (int rk = 0; rk < nr; ++rk)
                        if (hiBits == prefix[rk]) atomicAdd(&hist[rk][digit], 1u);
                }
            })
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
            double vi = __dmul_rn(__dsub_rn(ngood, 1.0), q);
            double t = __dsub_rn(vi, floor(vi));
            double a = unkey(prefix[2 * tid]), b = unkey(prefix[2 * tid + 1]);
            double diff = __dsub_rn(b, a);
            double lerp = __dadd_rn(a, __dmul_rn(diff, t));
            if (t >= 0.5) lerp = __dsub_rn(b, __dmul_rn(diff, __dsub_rn(1.0, t)));
            edges[tid] = lerp;
        }
        if (tid == 0) edges[nb] = wmax + 1e-6;
    }
    __syncthreads();
    if (tid <= nb) rec[1 + tid] = edges[tid];

    // ---- per-bin clipped standard deviation ----
    // With one bin every valid pixel belongs to it ([min, max + 1e-6) covers all valid weights): weights are not re-read.
#define NF_IN_BIN(val, q) ((val) != 0.0 && (ONEBIN || ((double)wts[q] >= lo && (double)wts[q] < hi)))
    for (int b = 0; b < nb; ++b) {
        const double lo = edges[b], hi = edges[b + 1];
        if (g.estimator == 1) {
            // RMSEstimator 'percentile': np.percentile(abs(binValues), 68.3) (nemo/filters.py:459-460), linear method:
            // the two order statistics around the virtual index by a radix select on the float64 keys of |v|
            double nn = 0.0;
            NF_CELL_LOOP(c, ww, wh, {
                double val = (double)v[q];
                if (NF_IN_BIN(val, q)) nn += 1.0;
            })
            const double n = block_sum(nn, red);
            if (n == 0.0) {              // percentile of an empty set is NaN: "chunkRMS > 0" fails
                if (tid == 0) rec[2 + nb + b] = 0.0;
                continue;
            }
            const double q683 = g.pctQ;
            const double vi = __dmul_rn(__dsub_rn(n, 1.0), q683);
            __syncthreads();
            if (tid < 2) {
                long long r = (long long)floor(vi) + tid;
                if (r > (long long)n - 1) r = (long long)n - 1;
                if (r < 0) r = 0;
                rankLeft[tid] = (unsigned int)r;
                prefix[tid] = 0ULL;
            }
            __syncthreads();
            for (int shift = 56; shift >= 0; shift -= 8) {
                for (int i = tid; i < 2 * 256; i += NF_NOISE_THREADS) (&hist[0][0])[i] = 0u;
                __syncthreads();
                NF_CELL_LOOP(c, ww, wh, {
                    double val = (double)v[q];
                    if (NF_IN_BIN(val, q)) {
                        unsigned long long key = key_of(fabs(val));
                        unsigned long long hiBits = (shift == 56) ? 0ULL : (key >> (shift + 8));
                        unsigned int digit = (unsigned int)((key >> shift) & 0xFFULL);
                        if (hiBits == prefix[0]) atomicAdd(&hist[0][digit], 1u);
                        if (hiBits == prefix[1]) atomicAdd(&hist[1][digit], 1u);
                    }
                })
                __syncthreads();
                if (tid < 2) {
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
            if (tid == 0) {
                const double t = __dsub_rn(vi, floor(vi));
                const double a = unkey(prefix[0]), bb = unkey(prefix[1]);
                const double diff = __dsub_rn(bb, a);
                double lerp = __dadd_rn(a, __dmul_rn(diff, t));
                if (t >= 0.5) lerp = __dsub_rn(bb, __dmul_rn(diff, __dsub_rn(1.0, t)));
                rec[2 + nb + b] = lerp > 0.0 ? lerp : 0.0;
            }
            __syncthreads();
            continue;
        }
        double thr = INFINITY;
        double mean = 0.0, rms = 0.0;
        bool have = false;
        for (int it = 0; it <= 10; ++it) {
            double m, r, n;
            if (it == 0) {
                // all valid pixels of the bin: two passes (mean, then squared deviations), as np.mean / np.std do
                double s = 0.0, nn = 0.0;
                NF_CELL_LOOP(c, ww, wh, {
                    double val = (double)v[q];
                    if (NF_IN_BIN(val, q)) {
                        s += val;
                        nn += 1.0;
                    }
                })
                s = block_sum(s, red);
                n = block_sum(nn, red);
                if (n == 0.0) break;      // empty bin -> rms 0
                m = s / n;
                double ss = 0.0;
                NF_CELL_LOOP(c, ww, wh, {
                    double val = (double)v[q];
                    if (NF_IN_BIN(val, q)) {
                        double d = val - m;
                        ss += d * d;
                    }
                })
                ss = block_sum(ss, red);
                r = sqrt(ss / n);
            } else {
                // clipped set |v| < |mean + 3 rms|: one pass of moments about the previous mean (which the new mean
                // differs from only by the clipped tail), so the variance suffers no cancellation
                const double c0 = mean;
                double s = 0.0, ss = 0.0, nn = 0.0;
                NF_CELL_LOOP(c, ww, wh, {
                    double val = (double)v[q];
                    if (fabs(val) < thr && NF_IN_BIN(val, q)) {
                        double d = val - c0;
                        s += d;
                        ss += d * d;
                        nn += 1.0;
                    }
                })
                s = block_sum(s, red);
                ss = block_sum(ss, red);
                n = block_sum(nn, red);
                if (n == 0.0) continue;   // "if mask.sum() > 0" fails: keep previous mean / rms, same threshold next time
                const double dm = s / n;
                m = c0 + dm;
                double var = ss / n - dm * dm;
                r = sqrt(var > 0.0 ? var : 0.0);
            }
            const bool same = have && (m == mean) && (r == rms);
            mean = m;
            rms = r;
            have = true;
            thr = fabs(mean + 3.0 * rms);
            if (same) break;          // fixed point: all remaining iterations reproduce these values
        }
        if (tid == 0) rec[2 + nb + b] = have ? rms : 0.0;
    }
#undef NF_IN_BIN
    if (tid == 0) rec[0] = 1.0;
}

// Register-resident variant: a 1024-thread workgroup holds the whole cell window (up to 1024 * NF_VPT pixels, e.g. the
// 160 x 160 window of a 40' grid at 0.5' pixels) in registers, so the map is read ONCE per cell (plus once per radix
// digit for the weights when there are several bins) and the <= 11 clipping iterations run out of registers.
#define NF_VPT 32
#define NF_REG_THREADS 1024

template <typename VT, typename WT, bool ONEBIN>
__global__ void __launch_bounds__(NF_REG_THREADS)
k_noise_cells_reg(const VT* __restrict__ v, const WT* __restrict__ wts, int H, int W, NoiseGrid g, double* __restrict__ stats) {
    const double* quantiles = g.quantiles;
    __shared__ double red[48];
    __shared__ unsigned int hist[ONEBIN ? 1 : 2 * NF_MAXBINS][256];
    __shared__ unsigned long long prefix[2 * NF_MAXBINS];
    __shared__ unsigned int rankLeft[2 * NF_MAXBINS];
    __shared__ double edges[NF_MAXBINS + 1];
    // blocks b, b + 8, ... share an XCD: give them consecutive cells, whose windows overlap by half (L2 reuse)
    const int ncell = g.ny * g.nx, per = (ncell + 7) >> 3;
    const int cell = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (cell >= ncell) return;
    const int nb = ONEBIN ? 1 : g.numBins;
    double* rec = stats + (long long)cell * cell_stride(nb);
    const CellWin c = cell_window(g, cell, H, W);
    const int ww = c.x1 - c.x0, wh = c.y1 - c.y0, npx = ww * wh;
    const int tid = threadIdx.x;

    double val[NF_VPT];
    unsigned long long binLo = 0ULL, binHi = 0ULL;   // 4-bit bin index of each held value
    unsigned int validMask = 0u;                       // bit j: held value j is a valid (non-zero) pixel
    double cnt = 0.0, wmin = INFINITY, wmax = -INFINITY;
    // element j of thread tid is window pixel j * 1024 + tid (consecutive lanes -> consecutive x)
    int r0 = tid / ww, x0 = tid - r0 * ww;
    const int dr = NF_REG_THREADS / ww, dx = NF_REG_THREADS - dr * ww;
    {
        // loads in branch-free groups of 8 (indices beyond the window are clamped to its first pixel and the value
        // discarded afterwards), so that 16 loads per thread are in flight instead of one dependent pair at a time
        int rr = r0, xx = x0;
        const long long qSafe = (long long)c.y0 * W + c.x0;
#pragma unroll
        for (int j0 = 0; j0 < NF_VPT; j0 += 8) {
            VT tv[8];
            WT tw[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const bool in = (j0 + jj) * NF_REG_THREADS + tid < npx;
                const long long q = in ? (long long)(c.y0 + rr) * W + c.x0 + xx : qSafe;
                tv[jj] = v[q];
                tw[jj] = wts[q];
                rr += dr;
                xx += dx;
                if (xx >= ww) {
                    xx -= ww;
                    rr += 1;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                const bool in = j * NF_REG_THREADS + tid < npx;
                const double t = in ? (double)tv[jj] : 0.0;
                if (t != 0.0) {
                    validMask |= 1u << j;
                    cnt += 1.0;
                    const double wv = (double)tw[jj];
                    wmin = fmin(wmin, wv);
                    wmax = fmax(wmax, wv);
                }
                val[j] = t;
            }
        }
    }
    const double ngood = block_sum(cnt, red);
    if (ngood == 0.0) {
        if (tid == 0) rec[0] = 0.0;
        return;
    }
    wmin = block_min(wmin, red);
    wmax = block_max(wmax, red);

    if (nb == 1) {
        if (tid == 0) {
            edges[0] = wmin;
            edges[1] = wmax + 1e-6;
        }
    } else if (!ONEBIN) {
        const int nr = 2 * nb;
        if (tid < nr) {
            double q = quantiles[tid / 2];
            double vi = __dmul_rn(__dsub_rn(ngood, 1.0), q);
            double fl = floor(vi);
            long long r = (long long)fl + (tid & 1);
            if (r > (long long)ngood - 1) r = (long long)ngood - 1;
            if (r < 0) r = 0;
            rankLeft[tid] = (unsigned int)r;
            prefix[tid] = 0ULL;
        }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int i = tid; i < nr * 256; i += NF_REG_THREADS) (&hist[0][0])[i] = 0u;
            __syncthreads();
            int rr = r0, xx = x0;
#pragma unroll 1
            for (int j = 0; j < NF_VPT; ++j) {
                if ((validMask >> j) & 1u) {
                    const long long q = (long long)(c.y0 + rr) * W + c.x0 + xx;
                    unsigned long long key = key_of((double)wts[q]);
                    unsigned long long hiBits = (shift == 56) ? 0ULL : (key >> (shift + 8));
                    unsigned int digit = (unsigned int)((key >> shift) & 0xFFULL);
                    for (int rk = 0; rk < nr; ++rk)
                        if (hiBits == prefix[rk]) atomicAdd(&hist[rk][digit], 1u);
                }
                rr += dr;
                xx += dx;
                if (xx >= ww) {
                    xx -= ww;
                    rr += 1;
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
            double vi = __dmul_rn(__dsub_rn(ngood, 1.0), q);
            double t = __dsub_rn(vi, floor(vi));
            double a = unkey(prefix[2 * tid]), b = unkey(prefix[2 * tid + 1]);
            double diff = __dsub_rn(b, a);
            double lerp = __dadd_rn(a, __dmul_rn(diff, t));
            if (t >= 0.5) lerp = __dsub_rn(b, __dmul_rn(diff, __dsub_rn(1.0, t)));
            edges[tid] = lerp;
        }
        if (tid == 0) edges[nb] = wmax + 1e-6;
    }
    __syncthreads();
    if (tid <= nb) rec[1 + tid] = edges[tid];

    if (!ONEBIN) {
        // bin of every held value: the LAST b with edges[b] <= w < edges[b+1] is irrelevant - bins are disjoint unless
        // edges repeat, and then the reference's later bins overwrite earlier ones only in the paint step (per-bin sets
        // are still "w >= lo and w < hi"), so a value may belong to no bin (15 = none) but never to two non-empty ones
        int rr = r0, xx = x0;
#pragma unroll 1
        for (int j = 0; j < NF_VPT; ++j) {
            unsigned long long bsel = 15ULL;
            if ((validMask >> j) & 1u) {
                const long long q = (long long)(c.y0 + rr) * W + c.x0 + xx;
                const double wv = (double)wts[q];
                for (int b = 0; b < nb; ++b)
                    if (wv >= edges[b] && wv < edges[b + 1]) bsel = (unsigned long long)b;
            }
            if (j < 16)
                binLo |= bsel << (4 * (j & 15));
            else
                binHi |= bsel << (4 * (j & 15));
            rr += dr;
            xx += dx;
            if (xx >= ww) {
                xx -= ww;
                rr += 1;
            }
        }
    }

#define NF_HELD_IN_BIN(j, b) (val[j] != 0.0 && (ONEBIN || (int)(((j) < 16 ? (binLo >> (4 * (j))) : (binHi >> (4 * ((j) - 16)))) & 15ULL) == (b)))
    for (int b = 0; b < nb; ++b) {
        double thr = INFINITY;
        double mean = 0.0, rms = 0.0;
        bool have = false;
        for (int it = 0; it <= 10; ++it) {
            double m, r, n;
            if (it == 0) {
                double s = 0.0, nn = 0.0;
#pragma unroll
                for (int j = 0; j < NF_VPT; ++j)
                    if (NF_HELD_IN_BIN(j, b)) {
                        s += val[j];
                        nn += 1.0;
                    }
                double zero = 0.0;
                block_sum3(s, nn, zero, red);
                n = nn;
                if (n == 0.0) break;
                m = s / n;
                double ss = 0.0;
#pragma unroll
                for (int j = 0; j < NF_VPT; ++j)
                    if (NF_HELD_IN_BIN(j, b)) {
                        double d = val[j] - m;
                        ss += d * d;
                    }
                ss = block_sum(ss, red);
                r = sqrt(ss / n);
            } else {
                const double c0 = mean;
                double s = 0.0, ss = 0.0, nn = 0.0;
#pragma unroll
                for (int j = 0; j < NF_VPT; ++j)
                    if (fabs(val[j]) < thr && NF_HELD_IN_BIN(j, b)) {
                        double d = val[j] - c0;
                        s += d;
                        ss += d * d;
                        nn += 1.0;
                    }
                block_sum3(s, ss, nn, red);
                n = nn;
                if (n == 0.0) continue;
                const double dm = s / n;
                m = c0 + dm;
                double var = ss / n - dm * dm;
                r = sqrt(var > 0.0 ? var : 0.0);
            }
            const bool same = have && (m == mean) && (r == rms);
            mean = m;
            rms = r;
            have = true;
            thr = fabs(mean + 3.0 * rms);
            if (same) break;
        }
        if (tid == 0) rec[2 + nb + b] = have ? rms : 0.0;
    }
#undef NF_HELD_IN_BIN
    if (tid == 0) rec[0] = 1.0;
}

// index range [lo, hi] of the cells whose widened window contains coordinate t along one axis (edges ascending)
__device__ __forceinline__ void cell_range(const int* __restrict__ e, int ncell, int overlap, int t, int* lo, int* hi) {
    // hi: last cell with e[i] - overlap <= t
    int a = 0, b = ncell - 1, h = -1;
    while (a <= b) {
        int mid = (a + b) >> 1;
        if (e[mid] - overlap <= t) {
            h = mid;
            a = mid + 1;
        } else
            b = mid - 1;
    }
    // lo: first cell with t < e[i+1] + overlap
    a = 0;
    b = ncell - 1;
    int l = ncell;
    while (a <= b) {
        int mid = (a + b) >> 1;
        if (t < e[mid + 1] + overlap) {
            l = mid;
            b = mid - 1;
        } else
            a = mid + 1;
    }
    *lo = l;
    *hi = h;
}

template <typename WT>
__global__ void k_noise_paint(const WT* __restrict__ wts, float* __restrict__ rmsMap, int H, int W, NoiseGrid g,
                              const double* __restrict__ stats) {
    const int nb = g.numBins;
    const int stride = cell_stride(nb);
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    int kLo, kHi;
    cell_range(g.xEdges, g.nx, g.overlap, x, &kLo, &kHi);
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        int iLo, iHi;
        cell_range(g.yEdges, g.ny, g.overlap, y, &iLo, &iHi);
        const long long p = (long long)y * W + x;
        const double wv = (double)wts[p];
        float outv = 0.0f;
        bool done = false;
        // the reference writes cells in (i, k, bin) order and later writes win: scan candidates backwards
        for (int i = iHi; i >= iLo && !done; --i) {
            for (int k = kHi; k >= kLo && !done; --k) {
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
    if (g.estimator != 0 && g.estimator != 1) {
        nf_set_error("noise_map: estimator must be 0 (3-sigma clipped standard deviation) or 1 (68.3 percentile of |v|)");
        return NF_ERR_ARG;
    }
    if (cellStatsBytes < noise_workspace_bytes(g)) {
        nf_set_error("noise_map: workspace too small");
        return NF_ERR_SIZE;
    }
    double* stats = (double*)cellStats;
    dim3 gc(g.ny * g.nx), bc(NF_NOISE_THREADS);
    // widest possible window: (largest cell + 2 * overlap)^2; cells differ by at most one pixel (np.linspace(dtype=int))
    const long long maxH = (long long)(H / g.ny + 1) + 2LL * g.overlap, maxW = (long long)(W / g.nx + 1) + 2LL * g.overlap;
    const bool inRegs = g.estimator == 0 && (maxH < H ? maxH : H) * (maxW < W ? maxW : W) <= (long long)NF_VPT * NF_REG_THREADS;
    if (inRegs) {
        bc = dim3(NF_REG_THREADS);
        gc = dim3(((g.ny * g.nx + 7) / 8) * 8);   // the register kernel permutes cells over the XCDs in groups of 8
    }
#define NF_NC(VT, WT)                                                                                                        \
    do {                                                                                                                     \
        if (inRegs && g.numBins == 1)                                                                                        \
            hipLaunchKernelGGL((k_noise_cells_reg<VT, WT, true>), gc, bc, 0, s, (const VT*)filtered, (const WT*)weights, H,  \
                               W, g, stats);                                                                                 \
        else if (inRegs)                                                                                                     \
            hipLaunchKernelGGL((k_noise_cells_reg<VT, WT, false>), gc, bc, 0, s, (const VT*)filtered, (const WT*)weights, H, \
                               W, g, stats);                                                                                 \
        else if (g.numBins == 1)                                                                                             \
            hipLaunchKernelGGL((k_noise_cells<VT, WT, true>), gc, bc, 0, s, (const VT*)filtered, (const WT*)weights, H, W, g, \
                               stats);                                                                                       \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_noise_cells<VT, WT, false>), gc, bc, 0, s, (const VT*)filtered, (const WT*)weights, H, W,  \
                               g, stats);                                                                                    \
    } while (0)
    if (vtype == NF_F64 && wtype == NF_F64) NF_NC(double, double);
    else if (vtype == NF_F64 && wtype == NF_F32) NF_NC(double, float);
    else if (vtype == NF_F32 && wtype == NF_F64) NF_NC(float, double);
    else NF_NC(float, float);
#undef NF_NC
    NF_CHECK_LAUNCH("k_noise_cells");
    dim3 gp(nf_div_up(W, 256), H < 1024 ? H : 1024);
    if (wtype == NF_F64)
        hipLaunchKernelGGL(k_noise_paint<double>, gp, dim3(256), 0, s, (const double*)weights, rms, H, W, g, stats);
    else
        hipLaunchKernelGGL(k_noise_paint<float>, gp, dim3(256), 0, s, (const float*)weights, rms, H, W, g, stats);
    NF_CHECK_LAUNCH("k_noise_paint");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// 'smart' mode (nemo/filters.py:363-404): noise measured over areas of similar weight, whole map at once.
//   binEdges = np.linspace(w.min(), w.max(), numBins); bin i = {binEdges[i] < w < binEdges[i+1]} (strict);
//   per bin the 3-sigma clipped standard deviation of the valid (non-zero) map values in it; RMSMap[bin] = rms if > 0.
// Every clipping iteration is one streaming pass over map + weights that accumulates (sum d, sum d^2, n) for all bins
// at once; per-workgroup partials are summed in a fixed order (deterministic), and the per-bin state (mean, rms,
// threshold) lives on the device, so the 13 passes are enqueued without a host round trip.
#define NF_SMART_BLOCKS 1024
#define NF_SMART_THREADS 256
#define NF_SMART_MAXB 15          // numNoiseBins <= 16 edges

struct SmartState {               // device-resident, one per call (inside the workspace)
    double wmin, wmax;
    double edges[NF_SMART_MAXB + 1];
    double mean[NF_SMART_MAXB], rms[NF_SMART_MAXB], thr[NF_SMART_MAXB];
    int have[NF_SMART_MAXB];
    int numEdges;
};

template <typename WT>
__global__ void __launch_bounds__(NF_SMART_THREADS) k_smart_minmax(const WT* __restrict__ w, long long n, double* __restrict__ part) {
    __shared__ double red[16];
    double lo = INFINITY, hi = -INFINITY;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const double x = (double)w[p];
        lo = fmin(lo, x);
        hi = fmax(hi, x);
    }
    lo = block_min(lo, red);
    __syncthreads();
    hi = block_max(hi, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = lo;
        part[2 * blockIdx.x + 1] = hi;
    }
}

__global__ void k_smart_edges(const double* __restrict__ part, int nPart, int numEdges, SmartState* st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int k = 0; k < nPart; ++k) {
        lo = fmin(lo, part[2 * k]);
        hi = fmax(hi, part[2 * k + 1]);
    }
    st->wmin = lo;
    st->wmax = hi;
    st->numEdges = numEdges;
    // np.linspace(start, stop, num): step = (stop - start) / (num - 1); y = arange(num) * step + start; y[-1] = stop
    const double step = numEdges > 1 ? __ddiv_rn(__dsub_rn(hi, lo), (double)(numEdges - 1)) : 0.0;
    for (int i = 0; i < numEdges; ++i) st->edges[i] = __dadd_rn(__dmul_rn((double)i, step), lo);
    if (numEdges > 1) st->edges[numEdges - 1] = hi;
    for (int b = 0; b < NF_SMART_MAXB; ++b) {
        st->mean[b] = 0.0;
        st->rms[b] = 0.0;
        st->thr[b] = INFINITY;
        st->have[b] = 0;
    }
}

__device__ __forceinline__ int smart_bin(const SmartState* st, double w, int nb) {
    for (int b = 0; b < nb; ++b)
        if (w > st->edges[b] && w < st->edges[b + 1]) return b;
    return -1;
}

// phase 0: sums of v (about 0) and counts over the valid pixels of each bin; phase 1: sums of (v - mean)^2;
// phase 2: clipped moments about the current mean.  part: [block][bin][3]
template <typename VT, typename WT>
__global__ void __launch_bounds__(NF_SMART_THREADS) k_smart_pass(const VT* __restrict__ v, const WT* __restrict__ w, long long n,
                                                                  const SmartState* __restrict__ st, int phase,
                                                                  double* __restrict__ part) {
    __shared__ double red[48];
    __shared__ SmartState ls;
    if (threadIdx.x == 0) ls = *st;
    __syncthreads();
    const int nb = ls.numEdges - 1;
    double s[NF_SMART_MAXB], ss[NF_SMART_MAXB], nn[NF_SMART_MAXB];
#pragma unroll
    for (int b = 0; b < NF_SMART_MAXB; ++b) s[b] = ss[b] = nn[b] = 0.0;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const double val = (double)v[p];
        if (val == 0.0) continue;
        const int bin = smart_bin(&ls, (double)w[p], nb);
        if (bin < 0) continue;
        double d = 0.0, take = 1.0;
        if (phase == 0) {
            d = val;
        } else if (phase == 1) {
            d = val - ls.mean[bin];
        } else {
            d = val - ls.mean[bin];
            if (!(fabs(val) < ls.thr[bin])) take = 0.0;
        }
        if (take != 0.0) {
#pragma unroll
            for (int b = 0; b < NF_SMART_MAXB; ++b)
                if (b == bin) {
                    s[b] += d;
                    ss[b] += d * d;
                    nn[b] += 1.0;
                }
        }
    }
    for (int b = 0; b < nb; ++b) {
        double a = s[b], q = ss[b], c = nn[b];
        block_sum3(a, q, c, red);
        if (threadIdx.x == 0) {
            double* o = part + ((long long)blockIdx.x * NF_SMART_MAXB + b) * 3;
            o[0] = a;
            o[1] = q;
            o[2] = c;
        }
        __syncthreads();
    }
}

// one thread per bin: fixed-order sum of the partials, then the state update of the reference's loop
__global__ void k_smart_update(const double* __restrict__ part, int nPart, int phase, SmartState* st) {
    const int b = threadIdx.x;
    if (b >= st->numEdges - 1) return;
    double s = 0.0, ss = 0.0, n = 0.0;
    for (int k = 0; k < nPart; ++k) {
        const double* o = part + ((long long)k * NF_SMART_MAXB + b) * 3;
        s += o[0];
        ss += o[1];
        n += o[2];
    }
    if (phase == 0) {
        st->have[b] = n > 0.0 ? 1 : 0;              // "np.not_equal(chunkValues, 0).sum() != 0"
        st->mean[b] = n > 0.0 ? s / n : 0.0;
    } else if (phase == 1) {
        if (st->have[b]) {
            st->rms[b] = sqrt(ss / n);
            st->thr[b] = fabs(st->mean[b] + 3.0 * st->rms[b]);
        }
    } else {
        if (st->have[b] && n > 0.0) {               // "if mask.sum() > 0"
            const double dm = s / n;
            const double var = ss / n - dm * dm;
            st->mean[b] = st->mean[b] + dm;
            st->rms[b] = sqrt(var > 0.0 ? var : 0.0);
            st->thr[b] = fabs(st->mean[b] + 3.0 * st->rms[b]);
        }
    }
}

template <typename WT>
__global__ void k_smart_paint(const WT* __restrict__ w, float* __restrict__ rmsMap, long long n, const SmartState* __restrict__ st) {
    __shared__ SmartState ls;
    if (threadIdx.x == 0) ls = *st;
    __syncthreads();
    const int nb = ls.numEdges - 1;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int bin = smart_bin(&ls, (double)w[p], nb);
        float out = 0.0f;
        if (bin >= 0 && ls.have[bin] && ls.rms[bin] > 0.0) out = (float)ls.rms[bin];
        rmsMap[p] = out;
    }
}

size_t noise_smart_workspace_bytes() {
    return sizeof(SmartState) + 256 + (size_t)NF_SMART_BLOCKS * NF_SMART_MAXB * 3 * sizeof(double);
}

int noise_map_smart(int vtype, const void* filtered, const void* weights, int wtype, float* rms, long long n, int numBins,
                    void* work, size_t workBytes, hipStream_t s) {
    if (numBins < 2 || numBins > NF_SMART_MAXB + 1) {
        nf_set_error("noise_map_smart: numNoiseBins must be in 2..16");
        return NF_ERR_ARG;
    }
    if (workBytes < noise_smart_workspace_bytes()) {
        nf_set_error("noise_map_smart: workspace too small");
        return NF_ERR_SIZE;
    }
    SmartState* st = (SmartState*)work;
    double* part = (double*)((char*)work + ((sizeof(SmartState) + 255) / 256) * 256);
    long long want = (n + NF_SMART_THREADS * 4LL - 1) / (NF_SMART_THREADS * 4LL);
    const int nPart = (int)(want < 1 ? 1 : (want > NF_SMART_BLOCKS ? NF_SMART_BLOCKS : want));
    dim3 g(nPart), b(NF_SMART_THREADS);
    if (wtype == NF_F64)
        hipLaunchKernelGGL(k_smart_minmax<double>, g, b, 0, s, (const double*)weights, n, part);
    else
        hipLaunchKernelGGL(k_smart_minmax<float>, g, b, 0, s, (const float*)weights, n, part);
    hipLaunchKernelGGL(k_smart_edges, dim3(1), dim3(1), 0, s, part, nPart, numBins, st);
    for (int it = 0; it < 12; ++it) {
        const int phase = it < 2 ? it : 2;
#define NF_SP(VT, WT) hipLaunchKernelGGL((k_smart_pass<VT, WT>), g, b, 0, s, (const VT*)filtered, (const WT*)weights, n, st, phase, part)
        if (vtype == NF_F64 && wtype == NF_F64) NF_SP(double, double);
        else if (vtype == NF_F64 && wtype == NF_F32) NF_SP(double, float);
        else if (vtype == NF_F32 && wtype == NF_F64) NF_SP(float, double);
        else NF_SP(float, float);
#undef NF_SP
        hipLaunchKernelGGL(k_smart_update, dim3(1), dim3(64), 0, s, part, nPart, phase, st);
    }
    if (wtype == NF_F64)
        hipLaunchKernelGGL(k_smart_paint<double>, g, b, 0, s, (const double*)weights, rms, n, st);
    else
        hipLaunchKernelGGL(k_smart_paint<float>, g, b, 0, s, (const float*)weights, rms, n, st);
    NF_CHECK_LAUNCH("noise_map_smart");
    return NF_OK;
}

}  // namespace nf

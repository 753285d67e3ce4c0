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

static bool env_flag_default_on(const char* name) {
    const char* e = getenv(name);
    return !(e && e[0] == '0');
}

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

// ---------------------------------------------------------------------------------------------------
// One-bin grid mode without re-reading the map ("block" path).
//
// The widened windows of neighbouring cells overlap (overlap = half a cell: every pixel lies in four windows), and the
// clipping loop visits every pixel of a window up to 11 times.  Here the map is read once:
//  k_noise_breaks   the window boundaries of all cells cut each axis into intervals (cut again where longer than
//                   NF_BLK); their products are the elementary blocks, and every window is a whole number of blocks.
//  k_noise_block_stats / k_noise_window_rms0 / k_noise_block_core
//                   one wavefront per block: count of valid pixels, sums, weight ranges; from those the unclipped rms of
//                   every window; then, with T = 2.25 x the smallest such rms among the block's windows, the CORE pixels
//                   |v| < T as three sums (n, sum (v - mu), sum (v - mu)^2, mu = block mean) and the TAIL pixels |v| >= T
//                   as a list of values (2 - 3 % of the pixels in Gaussian noise; the second sweep is served by the caches).
//  k_noise_windows  one workgroup per cell: the clipping iterations of nemo/filters.py:417-433 over its blocks.  An
//                   iteration with threshold thr >= T of a block takes all of the block's core pixels (|v| < T <= thr)
//                   from the three sums, shifted to the current centre, and tests only the tail list; a block with
//                   thr < T (a bright object inflated T, or much noisier than its window) is scanned pixel by pixel
//                   from the map for that iteration.  Same sums as the direct evaluation, regrouped (float64).
#define NF_BLK 48
#define NF_BLK_THREADS 256
#define NF_BLK_VPT (NF_BLK * NF_BLK / NF_BLK_THREADS)
#define NF_BREC 16   // per block: n valid, wmin, wmax, T, n core, mu, S1, S2, tail count, tail offset, wmin / wmax of all pixels,
                     // sum (v - v0), sum (v - v0)^2 over the valid pixels, v0

struct BlockWs {
    int* hdr;                      // [0] breakpoints along y, [1] along x
    int *breakY, *breakX;          // ascending pixel coordinates
    int *wY0, *wY1, *wX0, *wX1;    // per cell row / column: first block and one past the last block of its window
    double* rec;                   // [maxBy][maxBx][NF_BREC]
    double* tails;
    long long tailCap;             // room in `tails`
    // heavy blocks (more than NF_HEAVY tails - bright objects): their tails re-ordered by magnitude class in `tails2` (same
    // offsets as `tails`), per class (count, sum v, sum v^2, offset inside the list) in `cls`
    double* tails2;
    double* cls;                   // [maxHeavy][NF_NCLS][4]
    int* heavyIds;                 // [maxHeavy] block index by * maxBx + bx; hdr[4] = how many
    int maxHeavy;
    int maxBy, maxBx;
};

// Several maps of one grid and one weight map in one launch (the scales of a tile, nf_scale.hip): blockIdx.z is the map.  The maps
// lie vBytes apart, each has its own workspace (cell records and block tables) wsBytes after the one before, and its own output.
struct NoiseBatch {
    long long vBytes, wsBytes;
    float* rms[NF_SB_MAX];
};
__device__ __forceinline__ void ws_shift(BlockWs& ws, long long off) {
    if (off == 0) return;
#define NF_SH(f) ws.f = (decltype(ws.f))((char*)ws.f + off)
    NF_SH(hdr);
    NF_SH(breakY);
    NF_SH(breakX);
    NF_SH(wY0);
    NF_SH(wY1);
    NF_SH(wX0);
    NF_SH(wX1);
    NF_SH(rec);
    NF_SH(tails);
    NF_SH(tails2);
    NF_SH(cls);
    NF_SH(heavyIds);
#undef NF_SH
}
#define NF_NB_WS ws_shift(ws, (long long)blockIdx.z * nb.wsBytes)
#define NF_NB_V v = (decltype(v))((const char*)v + (long long)blockIdx.z * nb.vBytes)
#define NF_NB_STATS stats = (decltype(stats))((const char*)stats + (long long)blockIdx.z * nb.wsBytes)

#define NF_HEAVY 256
#define NF_NCLS 16
// lower bound of magnitude class c of a block with core threshold T: [T m_c, T m_(c+1)), the last class open above.  The
// final clipping threshold of a window lies between T and 2 T (T = 2.25 x the smallest typical block rms, threshold = 3 x
// the window's rms), so the classes are fine there (twelfth-octaves) and coarse (half-octaves) above.  One table for the
// kernel that sorts and the kernel that sums: the bounds are bitwise the same.
__device__ const double kClsMul[NF_NCLS + 1] = {1.0, 1.0594630943592953, 1.122462048309373, 1.189207115002721, 1.2599210498948732, 1.3348398541700344, 1.4142135623730951, 1.4983070768766815, 1.5874010519681994, 1.681792830507429, 1.7817974362806785, 1.8877486253633868, 2.0, 2.8284271247461903, 4.0, 5.656854249492381, INFINITY};
__device__ __forceinline__ double cls_bound(double T, int c) { return T * kClsMul[c]; }

static inline int blocks_max_breaks(int ncell, int len) { return 2 * ncell + len / NF_BLK + 5; }

__device__ void axis_breaks(const int* __restrict__ E, int n, int ov, int L, int* __restrict__ out, int maxOut, int* nOut) {
    // windows [a_i, b_i), a_i = max(E[i] - ov, 0), b_i = min(E[i+1] + ov, L): both sequences ascend; merge, drop repeats
    int ia = 0, ib = 0, cnt = 0, last = -1;
    if (max(E[0] - ov, 0) > 0) {
        out[cnt++] = 0;
        last = 0;
    }
    bool tail = false;
    while (ia < n || ib < n || !tail) {
        const int va = ia < n ? max(E[ia] - ov, 0) : 0x7fffffff;
        const int vb = ib < n ? min(E[ib + 1] + ov, L) : 0x7fffffff;
        int v = min(va, vb);
        if (ia >= n && ib >= n) {
            v = L;          // pixels after the last window: blocks that belong to no cell
            tail = true;
        } else if (va <= vb) ++ia; else ++ib;
        if (v == last) continue;
        if (last >= 0 && v - last > NF_BLK) {
            const int len = v - last, pieces = (len + NF_BLK - 1) / NF_BLK, step = (len + pieces - 1) / pieces;
            for (int j = 1; j < pieces; ++j)
                if (cnt < maxOut) out[cnt++] = last + j * step;
        }
        if (cnt < maxOut) out[cnt++] = v;
        last = v;
    }
    *nOut = cnt;
}

__device__ __forceinline__ int find_break(const int* __restrict__ b, int n, int v) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (b[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#define NF_BRK_MAXCELLS 1023
#define NF_BRK_MAXOUT 4096

// LDS form of axis_breaks for one axis: parallel merge of the two ascending sequences (ties: starts first), then one thread
// walks the merged list (repeats dropped, long intervals cut) - no dependent global loads anywhere.
__device__ void axis_merge_lds(const int* __restrict__ Eg, int n, int ov, int L, int* __restrict__ E, int* __restrict__ merged) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i <= n; i += nt) E[i] = Eg[i];
    __syncthreads();
    for (int i = tid; i < 2 * n; i += nt) {
        int v, pos;
        if (i < n) {            // a_i: after the b_j that are strictly smaller
            v = max(E[i] - ov, 0);
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (min(E[mid + 1] + ov, L) < v) lo = mid + 1; else hi = mid;
            }
            pos = i + lo;
        } else {                // b_j: after the a_i that are smaller or equal
            const int j = i - n;
            v = min(E[j + 1] + ov, L);
            int lo = 0, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (max(E[mid] - ov, 0) <= v) lo = mid + 1; else hi = mid;
            }
            pos = j + lo;
        }
        merged[pos] = v;
    }
}

// exclusive prefix over the 256 threads of the workgroup (wsum: 4 LDS ints); *total = sum over all threads
__device__ __forceinline__ int block_excl_scan256(int x, int* __restrict__ wsum, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = x;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = incl - x, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) before += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    return before;
}

// ext[0 .. m) = 0, the merged window boundaries, L (ascending).  Every distinct value is a breakpoint, and an interval
// longer than NF_BLK is cut into ceil(len / NF_BLK) pieces: element i contributes that many breakpoints (0 if it repeats
// its predecessor), written at the exclusive prefix of the counts.  Each thread takes a run of consecutive elements.
__device__ void axis_walk_lds(const int* __restrict__ ext, int m, int* __restrict__ out, int maxOut, int* __restrict__ wsum,
                              int* __restrict__ nOutLds) {
    const int per = (m + 255) / 256, i0 = threadIdx.x * per, i1 = min(i0 + per, m);
    int mine = 0;
    for (int i = i0; i < i1; ++i) {
        if (i == 0) mine += 1;
        else {
            const int len = ext[i] - ext[i - 1];
            mine += (len + NF_BLK - 1) / NF_BLK;      // 0 for a repeat
        }
    }
    int total;
    int pos = block_excl_scan256(mine, wsum, &total);
    for (int i = i0; i < i1; ++i) {
        if (i == 0) {
            if (pos < maxOut) out[pos] = ext[0];
            ++pos;
            continue;
        }
        const int prev = ext[i - 1], len = ext[i] - prev;
        if (len <= 0) continue;
        const int pieces = (len + NF_BLK - 1) / NF_BLK, step = (len + pieces - 1) / pieces;
        for (int j = 1; j < pieces; ++j, ++pos)
            if (pos < maxOut) out[pos] = prev + j * step;
        if (pos < maxOut) out[pos] = ext[i];
        ++pos;
    }
    if (threadIdx.x == 0) *nOutLds = min(total, maxOut);
}

__global__ void __launch_bounds__(256) k_noise_breaks(NoiseGrid g, int H, int W, BlockWs ws, NoiseBatch nb) {
    NF_NB_WS;
    __shared__ int E[2][NF_BRK_MAXCELLS + 1];
    __shared__ int merged[2][2 * NF_BRK_MAXCELLS + 2];   // 0, the merged boundaries, the axis length
    __shared__ int outL[2][NF_BRK_MAXOUT];
    __shared__ int nOut[2], wsum[4];
    const bool lds = g.ny <= NF_BRK_MAXCELLS && g.nx <= NF_BRK_MAXCELLS && ws.maxBy <= NF_BRK_MAXOUT && ws.maxBx <= NF_BRK_MAXOUT;
    const int* bY = ws.breakY;
    const int* bX = ws.breakX;
    if (lds) {
        axis_merge_lds(g.yEdges, g.ny, g.overlap, H, E[0], merged[0] + 1);
        axis_merge_lds(g.xEdges, g.nx, g.overlap, W, E[1], merged[1] + 1);
        if (threadIdx.x == 0) {
            merged[0][0] = 0;
            merged[0][2 * g.ny + 1] = H;
            merged[1][0] = 0;
            merged[1][2 * g.nx + 1] = W;
        }
        __syncthreads();
        axis_walk_lds(merged[0], 2 * g.ny + 2, outL[0], ws.maxBy, wsum, &nOut[0]);
        axis_walk_lds(merged[1], 2 * g.nx + 2, outL[1], ws.maxBx, wsum, &nOut[1]);
        __syncthreads();
        for (int i = threadIdx.x; i < nOut[0]; i += blockDim.x) ws.breakY[i] = outL[0][i];
        for (int i = threadIdx.x; i < nOut[1]; i += blockDim.x) ws.breakX[i] = outL[1][i];
        if (threadIdx.x == 0) {
            ws.hdr[0] = nOut[0];
            ws.hdr[1] = nOut[1];
            ws.hdr[2] = 0;
            ws.hdr[3] = 0;
            ws.hdr[4] = 0;
        }
        bY = outL[0];
        bX = outL[1];
    } else {
        if (threadIdx.x == 0) axis_breaks(g.yEdges, g.ny, g.overlap, H, ws.breakY, ws.maxBy, &nOut[0]);
        if (threadIdx.x == 64) axis_breaks(g.xEdges, g.nx, g.overlap, W, ws.breakX, ws.maxBx, &nOut[1]);
        __syncthreads();
        if (threadIdx.x == 0) {
            ws.hdr[0] = nOut[0];
            ws.hdr[1] = nOut[1];
            ws.hdr[2] = 0;
            ws.hdr[3] = 0;
            ws.hdr[4] = 0;
        }
    }
    const int nBy = nOut[0], nBx = nOut[1];
    for (int i = threadIdx.x; i < g.ny; i += blockDim.x) {
        ws.wY0[i] = find_break(bY, nBy, max(g.yEdges[i] - g.overlap, 0));
        ws.wY1[i] = find_break(bY, nBy, min(g.yEdges[i + 1] + g.overlap, H));
    }
    for (int k = threadIdx.x; k < g.nx; k += blockDim.x) {
        ws.wX0[k] = find_break(bX, nBx, max(g.xEdges[k] - g.overlap, 0));
        ws.wX1[k] = find_break(bX, nBx, min(g.xEdges[k + 1] + g.overlap, W));
    }
}

// Workgroup -> block: workgroups go to the 8 XCDs round robin, so XCD x takes the block rows x, x + 8, ... and walks each
// from left to right: the 128-byte lines that two neighbouring blocks share meet in one L2.
__device__ __forceinline__ bool block_of_workgroup(const BlockWs& ws, int* by, int* bx) {
    const int nby = ws.hdr[0] - 1, nbx = ws.hdr[1] - 1;
    const int xcd = (int)(blockIdx.x & 7u), idx = (int)(blockIdx.x >> 3);
    if (nbx < 1) return false;
    const int r = idx / nbx;
    *by = r * 8 + xcd;
    *bx = idx - r * nbx;
    return *by < nby;
}
static inline int blocks_grid(const BlockWs& ws) { return ((ws.maxBy - 1 + 7) / 8) * (ws.maxBx - 1) * 8; }

// Tail list room of block (r, c): the whole block (a block under a bright object can be all tails), so the lists tile a
// buffer of one value per pixel: offset = pixels of the block rows above + pixels of this row's blocks to the left.
__device__ __forceinline__ int tail_cap(int h, int w) { return h * w; }

__device__ __forceinline__ double wave_total(double v) {
    v = wave_sum_lane63(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ void cell_range(const int* __restrict__ e, int ncell, int overlap, int t, int* lo, int* hi);

// One WAVEFRONT per block (no barriers), lanes along x, then down the rows; loads in uniform groups of 4 per lane (slots
// past the block re-read its first pixel and are discarded).
// First sweep (k_noise_block_stats): valid count, sum v, sum v^2, weight ranges.
template <typename VT, typename WT>
__global__ void __launch_bounds__(64) k_noise_block_stats(const VT* __restrict__ v, const WT* __restrict__ wts, int W, BlockWs ws,
                                                         int reuse, NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_V;
    // reuse: the workspace still holds the block tables of an earlier call with the same grid and weights (k_noise_breaks did
    // not run: the call's counters are cleared here) and every block's weight range over all of its pixels (rec[10], rec[11]).
    // The weights are then read only for a block that is valid in part.
    if (reuse && blockIdx.x == 0 && threadIdx.x == 0) {
        ws.hdr[2] = 0;
        ws.hdr[3] = 0;
        ws.hdr[4] = 0;
    }
    int by, bx;
    if (!block_of_workgroup(ws, &by, &bx)) return;
    const int y0 = ws.breakY[by], y1 = ws.breakY[by + 1], x0 = ws.breakX[bx], x1 = ws.breakX[bx + 1];
    const int bw = x1 - x0, bh = y1 - y0, npx = bw * bh;
    const int lane = threadIdx.x;
    double* __restrict__ rec = ws.rec + ((long long)by * ws.maxBx + bx) * NF_BREC;
    const int r0 = lane / bw, c0 = lane - r0 * bw;
    const int dr = 64 / bw, dx = 64 - dr * bw;           // bw <= NF_BLK < 64
    const unsigned base = (unsigned)y0 * (unsigned)W + (unsigned)x0;   // 32-bit pixel offsets: H * W < 2^31 (host check)
    double sv = 0.0, sv2 = 0.0, n = 0.0, wmin = INFINITY, wmax = -INFINITY;
    double wminAll = INFINITY, wmaxAll = -INFINITY;     // over every pixel of the block, valid or not (for the paint step)
    // sums are taken about v0, the block's first pixel: a map with a large offset loses nothing to cancellation
    const double v0 = (double)v[base];
    const int iters = (npx + 63) >> 6;
    int rr = r0, xx = c0;
    for (int i0 = 0; i0 < iters; i0 += 4) {
        VT tv[4];
        WT tw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const unsigned q = in ? base + (unsigned)rr * (unsigned)W + (unsigned)xx : base;
            tv[u] = v[q];
            tw[u] = reuse ? (WT)0 : wts[q];
            rr += dr;
            xx += dx;
            if (xx >= bw) {
                xx -= bw;
                rr += 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const double t = in ? (double)tv[u] : 0.0;
            const double wv = (double)tw[u];
            if (in) {
                wminAll = fmin(wminAll, wv);
                wmaxAll = fmax(wmaxAll, wv);
            }
            if (t != 0.0) {
                const double d = t - v0;
                n += 1.0;
                sv += d;
                sv2 += d * d;
                wmin = fmin(wmin, wv);
                wmax = fmax(wmax, wv);
            }
        }
    }
    sv = wave_total(sv);
    sv2 = wave_total(sv2);
    n = wave_total(n);
    if (reuse) {
        wminAll = rec[10];
        wmaxAll = rec[11];
        wmin = n > 0.0 ? wminAll : INFINITY;
        wmax = n > 0.0 ? wmaxAll : -INFINITY;
        if (n > 0.0 && n < (double)npx) {
            // valid in part: the range over the valid pixels needs the weights after all (the map comes from the caches)
            wmin = INFINITY;
            wmax = -INFINITY;
            rr = r0;
            xx = c0;
            for (int i = 0; i < iters; ++i) {
                const bool in = i * 64 + lane < npx;
                const unsigned q = in ? base + (unsigned)rr * (unsigned)W + (unsigned)xx : base;
                const double t = in ? (double)v[q] : 0.0;
                const double wv = (double)wts[q];
                if (t != 0.0) {
                    wmin = fmin(wmin, wv);
                    wmax = fmax(wmax, wv);
                }
                rr += dr;
                xx += dx;
                if (xx >= bw) {
                    xx -= bw;
                    rr += 1;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        wmin = fmin(wmin, __shfl_xor(wmin, off, 64));
        wmax = fmax(wmax, __shfl_xor(wmax, off, 64));
        wminAll = fmin(wminAll, __shfl_xor(wminAll, off, 64));
        wmaxAll = fmax(wmaxAll, __shfl_xor(wmaxAll, off, 64));
    }
    if (lane == 0) {
        rec[0] = n;
        rec[1] = wmin;
        rec[2] = wmax;
        rec[8] = 0.0;
        rec[10] = wminAll;
        rec[11] = wmaxAll;
        rec[15] = -1.0;        // heavy slot (k_noise_block_core)
        rec[12] = sv;          // sum (v - v0) over the valid pixels
        rec[13] = sv2;         // sum (v - v0)^2
        rec[14] = v0;
    }
}

// A robust noise level for every cell's window from the block sums alone (one wavefront per cell, lane = block): the
// median of the blocks' unclipped standard deviations, weighted by their valid pixels - bright objects sit in a few blocks
// and leave it alone.  Left in the cell record's rms slot until k_noise_windows overwrites it; -1 for a window without
// valid pixels.  Windows of more than 64 blocks (a cell over a whole map) take the plain standard deviation of the window.
__global__ void __launch_bounds__(64) k_noise_window_rms0(NoiseGrid g, BlockWs ws, double* __restrict__ stats, NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_STATS;
    const int cell = blockIdx.x, lane = threadIdx.x;
    const int i = cell / g.nx, k = cell - i * g.nx;
    const int BY0 = ws.wY0[i], BX0 = ws.wX0[k], nbx = ws.wX1[k] - BX0, nblk = (ws.wY1[i] - BY0) * nbx;
    double r = -1.0;
    if (nblk <= 64) {
        double n = 0.0, rb_ = 0.0;
        if (lane < nblk) {
            const int byo = lane / nbx;
            const double* __restrict__ rb = ws.rec + ((long long)(BY0 + byo) * ws.maxBx + BX0 + (lane - byo * nbx)) * NF_BREC;
            n = rb[0];
            if (n > 0.0) {
                const double m = rb[12] / n, var = rb[13] / n - m * m;      // about the block's v0: the variance does not care
                rb_ = sqrt(var > 0.0 ? var : 0.0);
            }
        }
        double total = n;
        for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
        if (total > 0.0) {
            // weight of the blocks ordered before this one (ties by lane); the median block straddles half the total
            double before = 0.0;
            for (int j = 0; j < nblk; ++j) {
                const double nj = __shfl(n, j, 64), rj = __shfl(rb_, j, 64);
                if (rj < rb_ || (rj == rb_ && j < lane)) before += nj;
            }
            const bool med = n > 0.0 && before <= 0.5 * total && 0.5 * total < before + n;
            const unsigned long long m = __ballot(med);
            const int src = m ? __ffsll((long long)m) - 1 : 0;
            r = __shfl(rb_, src, 64);
        }
    } else {
        double n = 0.0, sv = 0.0, sv2 = 0.0;
        for (int bi = lane; bi < nblk; bi += 64) {
            const int byo = bi / nbx;
            const double* __restrict__ rb = ws.rec + ((long long)(BY0 + byo) * ws.maxBx + BX0 + (bi - byo * nbx)) * NF_BREC;
            // re-centre the block's sums on zero (this fallback serves whole-map cells; the robust median above is not
            // available for them)
            const double nb_ = rb[0], v0 = rb[14];
            n += nb_;
            sv += rb[12] + nb_ * v0;
            sv2 += rb[13] + 2.0 * v0 * rb[12] + nb_ * v0 * v0;
        }
        for (int off = 32; off > 0; off >>= 1) {
            n += __shfl_xor(n, off, 64);
            sv += __shfl_xor(sv, off, 64);
            sv2 += __shfl_xor(sv2, off, 64);
        }
        if (n > 0.0) {
            const double m = sv / n, var = sv2 / n - m * m;
            r = sqrt(var > 0.0 ? var : 0.0);
        }
    }
    if (lane == 0) stats[(long long)cell * cell_stride(1) + 3] = r;
}

// Second sweep (k_noise_block_core): T = 2.25 x the smallest of those noise levels among the windows that contain the
// block - the clipped rms of a window is rarely below 3/4 of its median block rms, so T stays under the windows' 3-sigma
// thresholds whatever the block itself looks like (correlated noise makes single blocks much noisier or quieter than their
// windows; a bright object fills its block's list instead of raising T).
// Core sums about the block mean, tail list in pixel order.  The block is found in the L2 / infinity cache.
template <typename VT>
__global__ void __launch_bounds__(64) k_noise_block_core(const VT* __restrict__ v, int W, NoiseGrid g, BlockWs ws,
                                                        const double* __restrict__ stats, NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_V;
    NF_NB_STATS;
    int by, bx;
    if (!block_of_workgroup(ws, &by, &bx)) return;
    const int y0 = ws.breakY[by], y1 = ws.breakY[by + 1], x0 = ws.breakX[bx], x1 = ws.breakX[bx + 1];
    const int bw = x1 - x0, bh = y1 - y0, npx = bw * bh;
    const int lane = threadIdx.x;
    double* __restrict__ rec = ws.rec + ((long long)by * ws.maxBx + bx) * NF_BREC;
    const double n = rec[0];
    if (n == 0.0) return;
    int iLo, iHi, kLo, kHi;
    cell_range(g.yEdges, g.ny, g.overlap, y0, &iLo, &iHi);
    cell_range(g.xEdges, g.nx, g.overlap, x0, &kLo, &kHi);
    const int nk = kHi - kLo + 1, ncand = (iHi >= iLo && nk > 0) ? (iHi - iLo + 1) * nk : 0;
    double rmin = INFINITY;
    for (int c = lane; c < ncand; c += 64) {
        const int ci = c / nk;
        const double r = stats[(long long)((iLo + ci) * g.nx + kLo + (c - ci * nk)) * cell_stride(1) + 3];
        if (r >= 0.0) rmin = fmin(rmin, r);
    }
    for (int off = 32; off > 0; off >>= 1) rmin = fmin(rmin, __shfl_xor(rmin, off, 64));
    if (ncand == 0) rmin = 0.0;                  // in no window: nothing will ask
    double T = 2.25 * rmin;
    const double mu = rec[14] + rec[12] / n;
    const int r0 = lane / bw, c0 = lane - r0 * bw;
    const int dr = 64 / bw, dx = 64 - dr * bw;
    const unsigned base = (unsigned)y0 * (unsigned)W + (unsigned)x0;
    const int cap = tail_cap(bh, bw);
    const long long tailOff = (long long)(y0 - ws.breakY[0]) * (ws.breakX[ws.hdr[1] - 1] - ws.breakX[0]) +
                              (long long)(x0 - ws.breakX[0]) * bh;
    const bool room = tailOff + cap <= ws.tailCap;
    double* __restrict__ tp = ws.tails + tailOff;
    double s1 = 0.0, s2 = 0.0, nc = 0.0;
    int total = 0;       // tails so far (uniform); list order = pixel order
    const int iters = (npx + 63) >> 6;
    int rr = r0, xx = c0;
    for (int i0 = 0; i0 < iters; i0 += 4) {
        VT tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const unsigned q = in ? base + (unsigned)rr * (unsigned)W + (unsigned)xx : base;
            tv[u] = v[q];
            rr += dr;
            xx += dx;
            if (xx >= bw) {
                xx -= bw;
                rr += 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const double t = in ? (double)tv[u] : 0.0;
            bool tail = false;
            if (t != 0.0) {
                if (fabs(t) < T) {
                    const double d = t - mu;
                    s1 += d;
                    s2 += d * d;
                    nc += 1.0;
                } else {
                    tail = true;
                }
            }
            const unsigned long long m = __ballot(tail);
            if (tail) {
                const int pos = total + __popcll(m & ((1ULL << lane) - 1ULL));
                if (room && pos < cap) tp[pos] = t;
            }
            total += __popcll(m);
        }
    }
    s1 = wave_total(s1);
    s2 = wave_total(s2);
    nc = wave_total(nc);
    if (!room || total > cap) {
        // no list room (cannot happen with a workspace of the documented size): the block is scanned directly in every
        // iteration (thr >= NaN is false)
        T = __longlong_as_double(0x7ff8000000000000LL);
        total = 0;
    }
    if (lane == 0) {
        int slot = -1;
        if (total > NF_HEAVY) {
            slot = atomicAdd(&ws.hdr[4], 1);
            if (slot < ws.maxHeavy) ws.heavyIds[slot] = by * ws.maxBx + bx; else slot = -1;
        }
        rec[3] = T;
        rec[4] = nc;
        rec[5] = mu;
        rec[6] = s1;
        rec[7] = s2;
        rec[8] = (double)total;
        rec[9] = (double)tailOff;
        rec[15] = (double)slot;
    }
}

// Heavy blocks: tails grouped by magnitude class (a counting sort by the 16 wavefronts of a workgroup; the order inside a
// class is fixed by the chunk -> wavefront assignment, so the result is the same in every run) with the class sums, so that
// a clipping iteration takes whole classes from their sums and tests the elements of one class only.
#define NF_CLS_WAVES 16
__device__ __forceinline__ int tail_class(bool in, double a, const double (&bound)[NF_NCLS]) {
    int k = in ? 0 : -1;
#pragma unroll
    for (int c = 1; c < NF_NCLS; ++c) k += (in && a >= bound[c]) ? 1 : 0;
    return k;
}

__global__ void __launch_bounds__(64 * NF_CLS_WAVES) k_noise_block_classes(BlockWs ws, NoiseBatch nb) {
    NF_NB_WS;
    __shared__ int cntW[NF_CLS_WAVES][NF_NCLS];      // per wavefront and class: count, then first position
    __shared__ int clsStart[NF_NCLS + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nHeavy = min(ws.hdr[4], ws.maxHeavy);
    for (int h = blockIdx.x; h < nHeavy; h += gridDim.x) {
        const double* __restrict__ rec = ws.rec + (long long)ws.heavyIds[h] * NF_BREC;
        const double T = rec[3];
        const int tc = (int)rec[8];
        const long long off = (long long)rec[9];
        const double* __restrict__ src = ws.tails + off;
        double* dst = ws.tails2 + off;
        double bound[NF_NCLS];
#pragma unroll
        for (int c = 0; c < NF_NCLS; ++c) bound[c] = cls_bound(T, c);
        // counts per wavefront and class (chunk i of 64 list entries belongs to wavefront i mod 16)
        int cnt[NF_NCLS];
#pragma unroll
        for (int c = 0; c < NF_NCLS; ++c) cnt[c] = 0;
        for (int e0 = w * 64; e0 < tc; e0 += 64 * NF_CLS_WAVES) {
            const int e = e0 + lane;
            const bool in = e < tc;
            const double t = src[in ? e : 0];
            const int k = tail_class(in, fabs(t), bound);
#pragma unroll
            for (int c = 0; c < NF_NCLS; ++c) cnt[c] += __popcll(__ballot(k == c));
        }
        if (lane < NF_NCLS) {
            int v = 0;
#pragma unroll
            for (int c = 0; c < NF_NCLS; ++c)
                if (c == lane) v = cnt[c];
            cntW[w][lane] = v;
        }
        __syncthreads();
        // first positions: classes one after the other, inside a class wavefront after wavefront
        if (threadIdx.x < 64) {
            int tot = 0;
            if (lane < NF_NCLS)
                for (int q = 0; q < NF_CLS_WAVES; ++q) tot += cntW[q][lane];
            int incl = tot;
            for (int o = 1; o < NF_NCLS; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (lane < NF_NCLS) {
                int run = incl - tot;
                clsStart[lane] = run;
                if (lane == NF_NCLS - 1) clsStart[NF_NCLS] = incl;
                for (int q = 0; q < NF_CLS_WAVES; ++q) {
                    const int n = cntW[q][lane];
                    cntW[q][lane] = run;
                    run += n;
                }
            }
        }
        __syncthreads();
        // placement
        int run[NF_NCLS];
#pragma unroll
        for (int c = 0; c < NF_NCLS; ++c) run[c] = cntW[w][c];
        for (int e0 = w * 64; e0 < tc; e0 += 64 * NF_CLS_WAVES) {
            const int e = e0 + lane;
            const bool in = e < tc;
            const double t = src[in ? e : 0];
            const int k = tail_class(in, fabs(t), bound);
#pragma unroll
            for (int c = 0; c < NF_NCLS; ++c) {
                const unsigned long long m = __ballot(k == c);
                if (k == c) dst[run[c] + __popcll(m & ((1ULL << lane) - 1ULL))] = t;
                run[c] += __popcll(m);
            }
        }
        __threadfence_block();
        __syncthreads();
        // class sums: one wavefront per class over its (now contiguous) segment
        for (int c = w; c < NF_NCLS; c += NF_CLS_WAVES) {
            const int s0 = clsStart[c], n = clsStart[c + 1] - s0;
            const double* seg = dst + s0;
            double s1 = 0.0, s2 = 0.0;
            for (int e0 = 0; e0 < n; e0 += 512) {
                double tv[8];                       // 8 loads per lane in flight
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * 64 + lane;
                    tv[u] = seg[e < n ? e : 0];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u * 64 + lane < n) {
                        s1 += tv[u];
                        s2 += tv[u] * tv[u];
                    }
            }
            s1 = wave_total(s1);
            s2 = wave_total(s2);
            if (lane == 0) {
                double* __restrict__ ce = ws.cls + ((long long)h * NF_NCLS + c) * 4;
                ce[0] = (double)n;
                ce[1] = s1;
                ce[2] = s2;
                ce[3] = (double)s0;
            }
        }
        __syncthreads();
    }
}

// ---- k_noise_windows: ONE WAVEFRONT per cell (no barriers; reductions on the vector ALU) ----
#define NF_WIN_MAXBLK 64      // blocks of a window whose records are staged in LDS (more: everything from global memory)
#define NF_WIN_HREG 4         // heavy blocks of a window whose class tables are held in registers (16 lanes each)
#define NF_WIN_MAXTAIL 1536   // tail values of a window held in LDS

// one block scanned pixel by pixel (its T exceeds the threshold of this iteration): flat over the block, 4 loads per lane
// in flight
template <typename VT>
__device__ __forceinline__ void scan_block(const VT* __restrict__ v, int W, const BlockWs& ws, int by, int bx, double c0, double thr,
                                           double& s, double& ss, double& nn) {
    const int lane = threadIdx.x & 63;
    const int y0 = ws.breakY[by], y1 = ws.breakY[by + 1], x0 = ws.breakX[bx], x1 = ws.breakX[bx + 1];
    const int bw = x1 - x0, npx = bw * (y1 - y0);
    const int dr = 64 / bw, dx = 64 - dr * bw;
    int rr = lane / bw, xx = lane - rr * bw;
    const unsigned base = (unsigned)y0 * (unsigned)W + (unsigned)x0;
    const int iters = (npx + 63) >> 6;
    for (int i0 = 0; i0 < iters; i0 += 4) {
        VT tv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const unsigned q = in ? base + (unsigned)rr * (unsigned)W + (unsigned)xx : base;
            tv[u] = v[q];
            rr += dr;
            xx += dx;
            if (xx >= bw) {
                xx -= bw;
                rr += 1;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = (i0 + u) * 64 + lane < npx;
            const double t = in ? (double)tv[u] : 0.0;
            if (t != 0.0 && fabs(t) < thr) {
                const double d = t - c0;
                s += d;
                ss += d * d;
                nn += 1.0;
            }
        }
    }
}

struct WinLds {
    int tailBase[NF_WIN_MAXBLK], tailCnt[NF_WIN_MAXBLK], heavy[NF_WIN_MAXBLK];   // heavy: class-table slot or -1
    long long tailOff[NF_WIN_MAXBLK];
    double tail[NF_WIN_MAXTAIL];
    unsigned char tailBlk[NF_WIN_MAXTAIL];
};

// the block that owns flat tail index e: binary search in the prefix of the tail counts
__device__ __forceinline__ int tail_block(const WinLds& L, int nblk, int e) {
    int lo = 0, hi = nblk - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (L.tailBase[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The tail values the LDS currently holds (the "working set"): all tails with |v| < thrW if `complete`.  Thresholds fall
// from one clipping iteration to the next, so after one pass over a long list the following iterations stay in LDS.
struct WinState {
    // lane = block: the block record (core threshold, core count/mean/sums about the mean; all valid pixels: count, sums
    // about v0) - registers, not LDS, which keeps ten windows resident per CU
    double T, nc, mu, S1, S2, nAll, v0, A1, A2;
    // the magnitude classes of the window's first NF_WIN_HREG heavy blocks: lane = 16 * (rank of the heavy block in the window)
    // + class (count, sum v, sum v^2, offset inside the block's class-ordered list, class bounds, block); hB < 0: nothing here.
    // hRest: the heavy blocks beyond those (bit = block), whose tables are read from global memory in every iteration.
    double hCn, hC1, hC2, hLo, hHi;
    int hOff, hB;
    unsigned long long hRest;
    // the straddling classes last read (hKey = their lanes, 0 = none) lie in LDS behind the working set, hE values: the
    // threshold moves little from one iteration to the next, so the same classes straddle it again
    unsigned long long hKey;
    int hE;
    int nW;
    double thrW;
    bool complete;
};

// sum d, sum d^2, count over the valid pixels of the window with |v| < thr, d = v - c0 (every lane gets the totals)
template <typename VT, bool LDS>
__device__ __forceinline__ void window_sums(const VT* __restrict__ v, int W, const BlockWs& ws, WinLds& L, WinState& st, int BY0,
                                            int BX0, int nbx, int nblk, int ntail, double c0, double thr, double& s, double& ss,
                                            double& nn) {
    const int lane = threadIdx.x & 63;
    s = 0.0;
    ss = 0.0;
    nn = 0.0;
    if (LDS) {
        // lane = block: core sums, or (rarely) a direct scan, which the whole wavefront does block after block
        bool scan = false, core = false;
        if (lane < nblk && st.nc >= 0.0) {
            core = thr >= st.T;
            if (core) {
                const double nc = st.nc, dmu = st.mu - c0, S1 = st.S1, S2 = st.S2;
                s = S1 + nc * dmu;
                ss = S2 + 2.0 * dmu * S1 + nc * dmu * dmu;
                nn = nc;
            } else {
                scan = true;
            }
        }
        const unsigned long long coreMask = __ballot(core);   // bit b: the tails of block b count (it is not scanned directly)
        if (thr == INFINITY) {
            // no clipping yet (iteration 0): every valid pixel counts - the blocks' all-pixel sums, no lists
            s = 0.0;
            ss = 0.0;
            nn = 0.0;
            if (lane < nblk && st.nc >= 0.0) {
                const double na = st.nAll, dv = st.v0 - c0, A1 = st.A1, A2 = st.A2;
                s = A1 + na * dv;
                ss = A2 + 2.0 * dv * A1 + na * dv * dv;
                nn = na;
            }
            scan = false;
        } else if (st.complete && thr <= st.thrW) {
#pragma unroll 4
            for (int e = lane; e < st.nW; e += 64) {
                const double t = L.tail[e];
                if (((coreMask >> L.tailBlk[e]) & 1ULL) && fabs(t) < thr) {
                    const double d = t - c0;
                    s += d;
                    ss += d * d;
                    nn += 1.0;
                }
            }
        } else {
            // one pass over the lists themselves, block after block (coalesced, 4 loads per lane in flight); the values
            // below thr become the new working set if they fit
            int nW = 0;
            bool fits = true;
            for (int b = 0; b < nblk; ++b) {
                const int tc = L.tailCnt[b];
                if (tc == 0) continue;
                const bool coreB = (coreMask >> b) & 1ULL;
                const double* __restrict__ tp = ws.tails + L.tailOff[b];
                for (int e0 = 0; e0 < tc; e0 += 512) {
                    double tv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * 64 + lane;
                        tv[u] = tp[e < tc ? e : 0];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * 64 + lane;
                        if (e0 + u * 64 >= tc) break;           // uniform
                        const double t = tv[u];
                        const bool below = e < tc && fabs(t) < thr;
                        if (below && coreB) {
                            const double d = t - c0;
                            s += d;
                            ss += d * d;
                            nn += 1.0;
                        }
                        const unsigned long long m = __ballot(below);
                        if (below && fits) {
                            const int pos = nW + __popcll(m & ((1ULL << lane) - 1ULL));
                            if (pos < NF_WIN_MAXTAIL) {
                                L.tail[pos] = t;
                                L.tailBlk[pos] = (unsigned char)b;
                            }
                        }
                        nW += __popcll(m);
                        if (nW > NF_WIN_MAXTAIL) fits = false;
                    }
                }
            }
            st.complete = fits;
            st.nW = fits ? nW : 0;
            st.thrW = thr;
            st.hKey = 0ULL;            // (the classes kept behind the old working set are gone)
            __syncthreads();
        }
        if (thr != INFINITY) {
            // heavy blocks in core mode: whole magnitude classes below thr from their sums, the one class that straddles thr
            // element by element.  The first NF_WIN_HREG heavy blocks of the window have their class tables in registers (no
            // load per iteration), and their straddling classes are read as ONE flattened list (independent loads).
            {
                const bool hv = st.hB >= 0 && ((coreMask >> (st.hB & 63)) & 1ULL) && st.hCn > 0.0;
                if (hv && st.hHi <= thr) {
                    const double cn = st.hCn, c1 = st.hC1, c2 = st.hC2;
                    s += c1 - cn * c0;
                    ss += c2 - 2.0 * c0 * c1 + cn * c0 * c0;
                    nn += cn;
                }
                const unsigned long long pm = __ballot(hv && st.hLo < thr && thr < st.hHi);
                if (pm && st.complete && pm == st.hKey) {
#pragma unroll 4
                    for (int e = lane; e < st.hE; e += 64) {
                        const double t = L.tail[st.nW + e];
                        if (fabs(t) < thr) {
                            const double d = t - c0;
                            s += d;
                            ss += d * d;
                            nn += 1.0;
                        }
                    }
                } else if (pm) {
                    int cntS[NF_WIN_HREG];
                    const double* ptrS[NF_WIN_HREG];
#pragma unroll
                    for (int h = 0; h < NF_WIN_HREG; ++h) {
                        const unsigned long long ph = pm & (0xFFFFULL << (16 * h));
                        cntS[h] = 0;
                        ptrS[h] = ws.tails2;
                        if (ph) {
                            const int srcLane = __ffsll((long long)ph) - 1;
                            cntS[h] = __shfl((int)st.hCn, srcLane, 64);
                            ptrS[h] = ws.tails2 + L.tailOff[__shfl(st.hB, srcLane, 64)] + __shfl(st.hOff, srcLane, 64);
                        }
                    }
                    const int e1 = cntS[0], e2 = e1 + cntS[1], e3 = e2 + cntS[2], E = e3 + cntS[3];
                    const bool keep = st.complete && E <= NF_WIN_MAXTAIL - st.nW;
                    for (int e0 = 0; e0 < E; e0 += 512) {
                        double tv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u * 64 + lane;
                            const double* p = e < e1 ? ptrS[0] + e : (e < e2 ? ptrS[1] + (e - e1) : (e < e3 ? ptrS[2] + (e - e2) : ptrS[3] + (e - e3)));
                            tv[u] = *(e < E ? p : ws.tails2);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u * 64 + lane;
                            if (keep && e < E) L.tail[st.nW + e] = tv[u];
                            if (e < E && fabs(tv[u]) < thr) {
                                const double d = tv[u] - c0;
                                s += d;
                                ss += d * d;
                                nn += 1.0;
                            }
                        }
                    }
                    st.hKey = keep ? pm : 0ULL;
                    st.hE = E;
                    __syncthreads();
                }
            }
            unsigned long long hm = __ballot(core && L.heavy[lane < nblk ? lane : 0] >= 0) & st.hRest;
            while (hm) {
                const int b = __ffsll((long long)hm) - 1;
                hm &= hm - 1;
                const double Tb = __shfl(st.T, b, 64);
                double cn = 0.0, c1 = 0.0, c2 = 0.0, lo = INFINITY, hi = INFINITY;
                int coff = 0;
                if (lane < NF_NCLS) {
                    const double* __restrict__ ce = ws.cls + ((long long)L.heavy[b] * NF_NCLS + lane) * 4;
                    cn = ce[0];
                    c1 = ce[1];
                    c2 = ce[2];
                    coff = (int)ce[3];
                    lo = cls_bound(Tb, lane);
                    hi = lane == NF_NCLS - 1 ? INFINITY : cls_bound(Tb, lane + 1);
                }
                if (cn > 0.0 && hi <= thr) {
                    s += c1 - cn * c0;
                    ss += c2 - 2.0 * c0 * c1 + cn * c0 * c0;
                    nn += cn;
                }
                const unsigned long long pm = __ballot(cn > 0.0 && lo < thr && thr < hi);
                if (pm) {
                    const int srcLane = __ffsll((long long)pm) - 1;
                    const int n_ = __shfl((int)cn, srcLane, 64), o_ = __shfl(coff, srcLane, 64);
                    const double* __restrict__ tp = ws.tails2 + L.tailOff[b] + o_;
                    for (int e0 = 0; e0 < n_; e0 += 512) {
                        double tv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u * 64 + lane;
                            tv[u] = tp[e < n_ ? e : 0];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int e = e0 + u * 64 + lane;
                            if (e < n_ && fabs(tv[u]) < thr) {
                                const double d = tv[u] - c0;
                                s += d;
                                ss += d * d;
                                nn += 1.0;
                            }
                        }
                    }
                }
            }
        }
        unsigned long long todo = __ballot(scan);
        if (todo && lane == 0) atomicAdd(&ws.hdr[2], __popcll(todo));    // diagnostics: direct block scans
        while (todo) {
            const int bi = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int byo = bi / nbx;
            scan_block<VT>(v, W, ws, BY0 + byo, BX0 + (bi - byo * nbx), c0, thr, s, ss, nn);
        }
    } else {
        // window too large for the LDS tables (a single cell over a whole map, say): lane = block, records and tail
        // lists straight from global memory
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int bi = b0 + lane;
            bool scan = false;
            if (bi < nblk) {
                const int byo = bi / nbx;
                const double* __restrict__ rb = ws.rec + ((long long)(BY0 + byo) * ws.maxBx + BX0 + (bi - byo * nbx)) * NF_BREC;
                if (rb[0] > 0.0) {
                    if (thr >= rb[3]) {
                        const double nc = rb[4], dmu = rb[5] - c0, S1 = rb[6], S2 = rb[7];
                        s += S1 + nc * dmu;
                        ss += S2 + 2.0 * dmu * S1 + nc * dmu * dmu;
                        nn += nc;
                        const int tc = (int)rb[8];
                        const double* __restrict__ tp = ws.tails + (long long)rb[9];
                        for (int e = 0; e < tc; ++e) {
                            const double t = tp[e];
                            if (fabs(t) < thr) {
                                const double d = t - c0;
                                s += d;
                                ss += d * d;
                                nn += 1.0;
                            }
                        }
                    } else {
                        scan = true;
                    }
                }
            }
            unsigned long long todo = __ballot(scan);
            while (todo) {
                const int bj = b0 + __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int byo = bj / nbx;
                scan_block<VT>(v, W, ws, BY0 + byo, BX0 + (bj - byo * nbx), c0, thr, s, ss, nn);
            }
        }
    }
    s = wave_total(s);
    ss = wave_total(ss);
    nn = wave_total(nn);
}

// 3-sigma clipping exactly as k_noise_cells_reg runs it (nemo/filters.py:417-433)
template <typename VT, bool LDS>
__device__ __forceinline__ double window_clip(const VT* __restrict__ v, int W, const BlockWs& ws, WinLds& L, WinState& st, int BY0,
                                              int BX0, int nbx, int nblk, int ntail) {
    double thr = INFINITY, mean = 0.0, rms = 0.0;
    bool have = false;
    for (int it = 0; it <= 10; ++it) {
        double m, r, s, ss, n;
        if (it == 0) {
            window_sums<VT, LDS>(v, W, ws, L, st, BY0, BX0, nbx, nblk, ntail, 0.0, INFINITY, s, ss, n);
            if (n == 0.0) break;
            m = s / n;
            window_sums<VT, LDS>(v, W, ws, L, st, BY0, BX0, nbx, nblk, ntail, m, INFINITY, s, ss, n);
            r = sqrt(ss / n);
        } else {
            const double c0 = mean;
            window_sums<VT, LDS>(v, W, ws, L, st, BY0, BX0, nbx, nblk, ntail, c0, thr, s, ss, n);
            if (n == 0.0) continue;
            const double dm = s / n;
            m = c0 + dm;
            const double var = ss / n - dm * dm;
            r = sqrt(var > 0.0 ? var : 0.0);
        }
        const bool same = have && (m == mean) && (r == rms);
        mean = m;
        rms = r;
        have = true;
        thr = fabs(mean + 3.0 * rms);
        if (same) break;
    }
    return have ? rms : 0.0;
}

template <typename VT>
__global__ void __launch_bounds__(64) k_noise_windows(const VT* __restrict__ v, int W, NoiseGrid g, BlockWs ws, double* __restrict__ stats,
                                                      NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_V;
    NF_NB_STATS;
    __shared__ WinLds L;
    const int cell = blockIdx.x, i = cell / g.nx, k = cell - i * g.nx;
    const int BY0 = ws.wY0[i], BX0 = ws.wX0[k], nbx = ws.wX1[k] - BX0, nblk = (ws.wY1[i] - BY0) * nbx;
    const int lane = threadIdx.x;
    double* __restrict__ rec = stats + (long long)cell * cell_stride(1);
    double cnt = 0.0, wmin = INFINITY, wmax = -INFINITY;
    int tcSum = 0;
    for (int bi = lane; bi < nblk; bi += 64) {
        const int byo = bi / nbx;
        const double* __restrict__ rb = ws.rec + ((long long)(BY0 + byo) * ws.maxBx + BX0 + (bi - byo * nbx)) * NF_BREC;
        const double nv = rb[0];
        if (nv > 0.0) {
            cnt += nv;
            wmin = fmin(wmin, rb[1]);
            wmax = fmax(wmax, rb[2]);
            if (rb[15] < 0.0) tcSum += (int)rb[8];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_xor(cnt, off, 64);
        wmin = fmin(wmin, __shfl_xor(wmin, off, 64));
        wmax = fmax(wmax, __shfl_xor(wmax, off, 64));
        tcSum += __shfl_xor(tcSum, off, 64);
    }
    if (cnt == 0.0) {
        if (lane == 0) rec[0] = 0.0;
        return;
    }
    double rms;
    if (nblk <= NF_WIN_MAXBLK) {
        // stage the block records (lane = block) and, through a prefix of the tail counts, the first NF_WIN_MAXTAIL tail
        // values (a window with more - bright objects - reads the rest from global memory in every iteration)
        WinState st;
        st.T = st.mu = st.S1 = st.S2 = st.nAll = st.A1 = st.A2 = st.v0 = 0.0;
        st.nc = -1.0;
        int tc = 0, slotR = -1;
        if (lane < nblk) {
            const int byo = lane / nbx;
            const double* __restrict__ rb = ws.rec + ((long long)(BY0 + byo) * ws.maxBx + BX0 + (lane - byo * nbx)) * NF_BREC;
            const bool live = rb[0] > 0.0;
            st.T = rb[3];
            st.nc = live ? rb[4] : -1.0;     // -1: block without valid pixels
            st.mu = rb[5];
            st.S1 = rb[6];
            st.S2 = rb[7];
            tc = live ? (int)rb[8] : 0;
            const int slot = live ? (int)rb[15] : -1;
            slotR = slot;
            L.heavy[lane] = slot;
            if (slot >= 0) tc = 0;           // a heavy block's tails are served from its class table, not from the LDS set
            L.tailOff[lane] = (long long)rb[9];
            L.tailCnt[lane] = tc;
            st.nAll = rb[0];
            st.A1 = rb[12];
            st.A2 = rb[13];
            st.v0 = rb[14];
        }
        int incl = tc;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (lane < nblk) L.tailBase[lane] = incl - tc;
        {
            // class tables of the first heavy blocks into registers, 16 lanes per block
            unsigned long long rest = __ballot(slotR >= 0);
            st.hB = -1;
            st.hOff = 0;
            st.hCn = st.hC1 = st.hC2 = 0.0;
            st.hLo = st.hHi = INFINITY;
#pragma unroll
            for (int h = 0; h < NF_WIN_HREG; ++h) {
                if (!rest) break;
                const int b = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const int slotB = __shfl(slotR, b, 64);
                const double Tb = __shfl(st.T, b, 64);
                if ((lane >> 4) == h) {
                    const int c = lane & 15;
                    const double* __restrict__ ce = ws.cls + ((long long)slotB * NF_NCLS + c) * 4;
                    st.hCn = ce[0];
                    st.hC1 = ce[1];
                    st.hC2 = ce[2];
                    st.hOff = (int)ce[3];
                    st.hLo = cls_bound(Tb, c);
                    st.hHi = c == NF_NCLS - 1 ? INFINITY : cls_bound(Tb, c + 1);
                    st.hB = b;
                }
            }
            st.hRest = rest;
            st.hKey = 0ULL;
            st.hE = 0;
        }
        __syncthreads();
        // the block of flat tail index e by binary search in the prefix (LDS only), then the loads, all independent
        st.complete = tcSum <= NF_WIN_MAXTAIL;
        st.nW = st.complete ? tcSum : 0;
        st.thrW = st.complete ? INFINITY : 0.0;
        if (st.complete) {
            for (int e = lane; e < tcSum; e += 64) L.tailBlk[e] = (unsigned char)tail_block(L, nblk, e);
#pragma unroll 8
            for (int e = lane; e < tcSum; e += 64) {
                const int bl = L.tailBlk[e];
                L.tail[e] = ws.tails[L.tailOff[bl] + (e - L.tailBase[bl])];
            }
        } else if (lane == 0) {
            atomicAdd(&ws.hdr[3], 1);        // diagnostics: windows with more tails than LDS room
        }
        __syncthreads();
        rms = window_clip<VT, true>(v, W, ws, L, st, BY0, BX0, nbx, nblk, tcSum);
    } else {
        WinState st;
        st.T = st.mu = st.S1 = st.S2 = st.nAll = st.A1 = st.A2 = st.v0 = 0.0;
        st.nc = -1.0;
        st.complete = false;
        st.nW = 0;
        st.thrW = 0.0;
        st.hB = -1;
        st.hOff = 0;
        st.hCn = st.hC1 = st.hC2 = 0.0;
        st.hLo = st.hHi = INFINITY;
        st.hRest = 0;
        st.hKey = 0ULL;
        st.hE = 0;
        rms = window_clip<VT, false>(v, W, ws, L, st, BY0, BX0, nbx, nblk, 0);
    }
    if (lane == 0) {
        rec[0] = 1.0;
        rec[1] = wmin;
        rec[2] = wmax + 1e-6;
        rec[3] = rms;
    }
}

// Paint for the block path: all pixels of a block lie in the same windows, so the candidate cells (in the reference's
// overwrite order, last writer first; only active cells with rms > 0) are found once per block.
#define NF_PAINT_MAXCAND 32
template <typename WT>
__global__ void __launch_bounds__(NF_BLK_THREADS)
k_noise_paint_blocks(const WT* __restrict__ wts, float* __restrict__ rmsMap, int W, NoiseGrid g, BlockWs ws,
                     const double* __restrict__ stats, NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_STATS;
    rmsMap = nb.rms[blockIdx.z];
    __shared__ double cLo[NF_PAINT_MAXCAND], cHi[NF_PAINT_MAXCAND];
    __shared__ float cR[NF_PAINT_MAXCAND];
    __shared__ int nCand, range[4];
    int by, bx;
    if (!block_of_workgroup(ws, &by, &bx)) return;
    const int y0 = ws.breakY[by], y1 = ws.breakY[by + 1], x0 = ws.breakX[bx], x1 = ws.breakX[bx + 1];
    const int bw = x1 - x0, bh = y1 - y0, npx = bw * bh;
    const int tid = threadIdx.x;
    // the first wavefront looks at the candidate cells side by side (lane = candidate, in overwrite order) and keeps the
    // usable ones in that order: one round of loads instead of a chain of dependent ones
    const double* __restrict__ rb = ws.rec + ((long long)by * ws.maxBx + bx) * NF_BREC;
    const double wLoAll = rb[10], wHiAll = rb[11];
    if (tid < 64) {
        int iLo, iHi, kLo, kHi;
        cell_range(g.yEdges, g.ny, g.overlap, y0, &iLo, &iHi);
        cell_range(g.xEdges, g.nx, g.overlap, x0, &kLo, &kHi);
        const int nk = kHi - kLo + 1, ni = iHi - iLo + 1, ncand = (ni > 0 && nk > 0) ? ni * nk : 0;
        int n = 0;
        for (int c0 = 0; c0 < ncand; c0 += 64) {
            const int c = c0 + tid;
            bool ok = false;
            double lo = 0.0, hi = 0.0, r = 0.0;
            if (c < ncand) {
                const int ci = c / nk;
                const double* rec = stats + (long long)((iHi - ci) * g.nx + kHi - (c - ci * nk)) * cell_stride(1);
                const double act = rec[0];
                lo = rec[1];
                hi = rec[2];
                r = rec[3];
                ok = act != 0.0 && r > 0.0;
            }
            const unsigned long long m = __ballot(ok);
            const int pos = n + __popcll(m & ((1ULL << tid) - 1ULL));
            if (ok && pos < NF_PAINT_MAXCAND) {
                cLo[pos] = lo;
                cHi[pos] = hi;
                cR[pos] = (float)r;
            }
            n += __popcll(m);
        }
        if (tid == 0) {
            nCand = n > NF_PAINT_MAXCAND ? -1 : n;
            range[0] = iLo;
            range[1] = iHi;
            range[2] = kLo;
            range[3] = kHi;
        }
    }
    __syncthreads();
    // every weight of the block inside the first candidate's range: one value for the whole block, weights not read
    const int nc = nCand;
    int rr = tid / bw, xx = tid - rr * bw;
    const int dr = NF_BLK_THREADS / bw, dx = NF_BLK_THREADS - dr * bw;
    const bool uni = nc >= 1 && wLoAll >= cLo[0] && wHiAll < cHi[0];
    const float uniVal = uni ? cR[0] : 0.0f;
    for (int p = tid; p < npx; p += NF_BLK_THREADS) {
        const long long q = (long long)(y0 + rr) * W + x0 + xx;
        float outv = 0.0f;
        const double wv = uni ? 0.0 : (double)wts[q];
        if (uni) {
            outv = uniVal;
        } else if (nc >= 0) {
            for (int c = 0; c < nc; ++c)
                if (wv >= cLo[c] && wv < cHi[c]) {
                    outv = cR[c];
                    break;
                }
        } else {
            bool done = false;
            for (int i = range[1]; i >= range[0] && !done; --i)
                for (int k = range[3]; k >= range[2] && !done; --k) {
                    const double* rec = stats + (long long)(i * g.nx + k) * cell_stride(1);
                    if (rec[0] == 0.0) continue;
                    if (wv >= rec[1] && wv < rec[2] && rec[3] > 0.0) {
                        outv = (float)rec[3];
                        done = true;
                    }
                }
        }
        rmsMap[q] = outv;
        rr += dr;
        xx += dx;
        if (xx >= bw) {
            xx -= bw;
            rr += 1;
        }
    }
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

// Paint by rows: one workgroup per (block row, group of NF_PR_ROWS rows).  Per block of the row, the value every pixel takes when
// all of the block's weights lie inside the range of its last-writing usable cell (the common case: weights are then not
// read) - found once, lane = block; the rows then leave as whole lines, 16 bytes per lane.  A block whose weights straddle a
// range boundary is painted pixel by pixel like k_noise_paint does it.
#define NF_PR_ROWS 8
#define NF_PR_MAXBX 1024
template <typename WT>
__device__ float paint_pixel(const WT* __restrict__ wts, long long q, const NoiseGrid& g, const double* __restrict__ stats, int iLo,
                             int iHi, int kLo, int kHi) {
    const double wv = (double)wts[q];
    for (int i = iHi; i >= iLo; --i)
        for (int k = kHi; k >= kLo; --k) {
            const double* rec = stats + (long long)(i * g.nx + k) * cell_stride(1);
            if (rec[0] == 0.0) continue;
            if (wv >= rec[1] && wv < rec[2] && rec[3] > 0.0) return (float)rec[3];
        }
    return 0.0f;
}

template <typename WT>
__global__ void __launch_bounds__(256)
k_noise_paint_rows(const WT* __restrict__ wts, float* __restrict__ rmsMap, int W, NoiseGrid g, BlockWs ws,
                   const double* __restrict__ stats, NoiseBatch nb) {
    NF_NB_WS;
    NF_NB_STATS;
    rmsMap = nb.rms[blockIdx.z];
    __shared__ int bX[NF_PR_MAXBX + 1];
    __shared__ float val[NF_PR_MAXBX];
    __shared__ unsigned char uni[NF_PR_MAXBX];
    const int nby = ws.hdr[0] - 1, nbx = ws.hdr[1] - 1;
    const int by = blockIdx.y;
    if (by >= nby || nbx < 1) return;
    const int y0 = ws.breakY[by], y1 = ws.breakY[by + 1];
    const int ya = y0 + (int)blockIdx.x * NF_PR_ROWS, yb = min(ya + NF_PR_ROWS, y1);
    if (ya >= y1) return;
    const int tid = threadIdx.x;
    int iLo, iHi;
    cell_range(g.yEdges, g.ny, g.overlap, y0, &iLo, &iHi);
    for (int bx = tid; bx <= nbx; bx += 256) bX[bx] = ws.breakX[bx];
    for (int bx = tid; bx < nbx; bx += 256) {
        int kLo, kHi;
        cell_range(g.xEdges, g.nx, g.overlap, ws.breakX[bx], &kLo, &kHi);
        // the last writer among the usable cells (nemo/filters.py:470-478 overwrites in (i, k) order)
        bool found = false;
        double lo = 0.0, hi = 0.0, r = 0.0;
        for (int i = iHi; i >= iLo && !found; --i)
            for (int k = kHi; k >= kLo && !found; --k) {
                const double* rec = stats + (long long)(i * g.nx + k) * cell_stride(1);
                if (rec[0] == 0.0 || !(rec[3] > 0.0)) continue;
                lo = rec[1];
                hi = rec[2];
                r = rec[3];
                found = true;
            }
        const double* rb = ws.rec + ((long long)by * ws.maxBx + bx) * NF_BREC;
        const bool u = !found || (rb[10] >= lo && rb[11] < hi);
        uni[bx] = u ? 1 : 0;
        val[bx] = found ? (float)r : 0.0f;
    }
    __syncthreads();
    const int x0 = bX[0], xEnd = bX[nbx];          // (the blocks cover the whole map: 0 .. W)
    for (int xa = x0 + tid * 4; xa < xEnd; xa += 1024) {
        // block of xa, then forward through the boundaries for the next three pixels
        int lo = 0, hi = nbx - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (bX[mid] <= xa) lo = mid; else hi = mid - 1;
        }
        int bxs[4];
        bool allUni = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            while (lo < nbx - 1 && bX[lo + 1] <= xa + j) ++lo;
            bxs[j] = lo;
            allUni = allUni && uni[lo] != 0;
        }
        const bool vec = allUni && xa + 3 < xEnd && (W & 3) == 0 && (x0 & 3) == 0 && ((size_t)rmsMap & 15) == 0;
        if (vec) {
            float4 o;
            o.x = val[bxs[0]];
            o.y = val[bxs[1]];
            o.z = val[bxs[2]];
            o.w = val[bxs[3]];
            for (int y = ya; y < yb; ++y) *(float4*)(rmsMap + (long long)y * W + xa) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xa + j;
                if (x >= xEnd) break;
                const int bx = bxs[j];
                if (uni[bx]) {
                    const float o = val[bx];
                    for (int y = ya; y < yb; ++y) rmsMap[(long long)y * W + x] = o;
                } else {
                    int kLo, kHi;
                    cell_range(g.xEdges, g.nx, g.overlap, bX[bx], &kLo, &kHi);
                    for (int y = ya; y < yb; ++y) {
                        const long long q = (long long)y * W + x;
                        rmsMap[q] = paint_pixel<WT>(wts, q, g, stats, iLo, iHi, kLo, kHi);
                    }
                }
            }
        }
    }
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

static size_t align8(size_t b) { return (b + 7) & ~(size_t)7; }
static size_t noise_max_heavy(size_t maxBy, size_t maxBx) { return maxBy * maxBx < 16384 ? maxBy * maxBx : 16384; }
// sum of tail_cap over any block decomposition of an H x W map with at most maxBy x maxBx blocks
static size_t noise_tail_cap(int H, int W, int maxBy, int maxBx) {
    (void)maxBy;
    (void)maxBx;
    return (size_t)H * (size_t)W + 8;
}

// cell records, then (one-bin clipped-RMS mode) the block tables: see BlockWs
size_t noise_workspace_bytes(const NoiseGrid& g, int H, int W) {
    size_t cells = (size_t)g.ny * g.nx * cell_stride(g.numBins) * sizeof(double);
    if (g.numBins != 1) return cells;
    const size_t maxBy = blocks_max_breaks(g.ny, H), maxBx = blocks_max_breaks(g.nx, W);
    size_t ints = align8((8 + maxBy + maxBx + 2 * (size_t)g.ny + 2 * (size_t)g.nx) * sizeof(int));
    size_t recs = maxBy * maxBx * NF_BREC * sizeof(double);
    size_t tails = noise_tail_cap(H, W, (int)maxBy, (int)maxBx) * sizeof(double);
    size_t heavy = align8(noise_max_heavy(maxBy, maxBx) * sizeof(int)) + noise_max_heavy(maxBy, maxBx) * NF_NCLS * 4 * sizeof(double);
    return cells + ints + recs + 2 * tails + heavy;
}

int noise_map(int vtype, const void* filtered, const void* weights, int wtype, float* rms, int H, int W, const NoiseGrid& g,
              void* cellStats, size_t cellStatsBytes, bool tablesValid, hipStream_t s) {
    float* one[1] = {rms};
    return noise_map_batch(vtype, filtered, 0, weights, wtype, one, 1, 1, H, W, g, cellStats, cellStatsBytes, 0, tablesValid, s);
}

bool noise_batchable(const NoiseGrid& g) {
    static const bool blockPath = env_flag_default_on("NF_NOISE_BLOCKS");
    return g.estimator == 0 && g.numBins == 1 && blockPath;
}

// nb maps (vStrideBytes apart) of one grid and one weight map; map b uses the workspace cellStats + b * wsStrideBytes (each of
// cellStatsBytes) and writes rms[b].  When the block tables are not valid they are made for nbTables >= nb workspaces, so that
// "valid" afterwards holds for all of them.  nb > 1 needs noise_batchable(g).
int noise_map_batch(int vtype, const void* filtered, long long vStrideBytes, const void* weights, int wtype, float* const* rmsList, int nb,
                    int nbTables, int H, int W, const NoiseGrid& g, void* cellStats, size_t cellStatsBytes, long long wsStrideBytes,
                    bool tablesValid, hipStream_t s) {
    if (nb < 1 || nb > NF_SB_MAX || nbTables < nb || (nb > 1 && (!noise_batchable(g) || (wsStrideBytes & 7) != 0))) {
        nf_set_error("noise_map_batch: 1..8 maps per launch (one-bin clipped-RMS grid mode for more than one)");
        return NF_ERR_ARG;
    }
    float* rms = rmsList[0];
    NoiseBatch nbArgs;
    memset(&nbArgs, 0, sizeof(nbArgs));
    nbArgs.vBytes = vStrideBytes;
    nbArgs.wsBytes = nb > 1 || nbTables > 1 ? wsStrideBytes : 0;
    for (int b = 0; b < nb; ++b) nbArgs.rms[b] = rmsList[b];
    if (g.numBins < 1 || g.numBins > NF_MAXBINS) {
        nf_set_error("noise_map: numNoiseBins must be in 1..16");
        return NF_ERR_ARG;
    }
    if (g.estimator != 0 && g.estimator != 1) {
        nf_set_error("noise_map: estimator must be 0 (3-sigma clipped standard deviation) or 1 (68.3 percentile of |v|)");
        return NF_ERR_ARG;
    }
    if (cellStatsBytes < noise_workspace_bytes(g, H, W)) {
        nf_set_error("noise_map: workspace too small");
        return NF_ERR_SIZE;
    }
    double* stats = (double*)cellStats;
    static const bool blockPath = env_flag_default_on("NF_NOISE_BLOCKS");
    if (g.estimator == 0 && g.numBins == 1 && blockPath) {
        BlockWs ws;
        ws.maxBy = blocks_max_breaks(g.ny, H);
        ws.maxBx = blocks_max_breaks(g.nx, W);
        char* p = (char*)cellStats + (size_t)g.ny * g.nx * cell_stride(1) * sizeof(double);
        ws.hdr = (int*)p;
        ws.tailCap = (long long)noise_tail_cap(H, W, ws.maxBy, ws.maxBx);
        ws.breakY = ws.hdr + 8;
        ws.breakX = ws.breakY + ws.maxBy;
        ws.wY0 = ws.breakX + ws.maxBx;
        ws.wY1 = ws.wY0 + g.ny;
        ws.wX0 = ws.wY1 + g.ny;
        ws.wX1 = ws.wX0 + g.nx;
        p += align8((8 + (size_t)ws.maxBy + (size_t)ws.maxBx + 2 * (size_t)g.ny + 2 * (size_t)g.nx) * sizeof(int));
        ws.rec = (double*)p;
        ws.tails = ws.rec + (size_t)ws.maxBy * ws.maxBx * NF_BREC;
        ws.tails2 = ws.tails + noise_tail_cap(H, W, ws.maxBy, ws.maxBx);
        ws.maxHeavy = (int)noise_max_heavy(ws.maxBy, ws.maxBx);
        ws.cls = ws.tails2 + noise_tail_cap(H, W, ws.maxBy, ws.maxBx);
        ws.heavyIds = (int*)(ws.cls + (size_t)ws.maxHeavy * NF_NCLS * 4);
        if ((long long)H * W >= (1LL << 31)) {
            nf_set_error("noise_map: map too large for the 32-bit pixel offsets of the block kernels");
            return NF_ERR_SIZE;
        }
        const int reuse = tablesValid ? 1 : 0;
        if (!reuse) {
            hipLaunchKernelGGL(k_noise_breaks, dim3(1, 1, nbTables), dim3(256), 0, s, g, H, W, ws, nbArgs);
            NF_CHECK_LAUNCH("k_noise_breaks");
        }
        dim3 gb(blocks_grid(ws), 1, nb), gw(g.ny * g.nx, 1, nb);
#define NF_NB(VT, WT)                                                                                                      \
    do {                                                                                                                   \
        hipLaunchKernelGGL((k_noise_block_stats<VT, WT>), gb, dim3(64), 0, s, (const VT*)filtered, (const WT*)weights, W, ws, reuse, nbArgs); \
        hipLaunchKernelGGL(k_noise_window_rms0, gw, dim3(64), 0, s, g, ws, stats, nbArgs);                                 \
        hipLaunchKernelGGL((k_noise_block_core<VT>), gb, dim3(64), 0, s, (const VT*)filtered, W, g, ws, stats, nbArgs);    \
        hipLaunchKernelGGL(k_noise_block_classes, dim3(512, 1, nb), dim3(64 * NF_CLS_WAVES), 0, s, ws, nbArgs);            \
        hipLaunchKernelGGL((k_noise_windows<VT>), gw, dim3(64), 0, s, (const VT*)filtered, W, g, ws, stats, nbArgs);       \
    } while (0)
        if (vtype == NF_F64 && wtype == NF_F64) NF_NB(double, double);
        else if (vtype == NF_F64 && wtype == NF_F32) NF_NB(double, float);
        else if (vtype == NF_F32 && wtype == NF_F64) NF_NB(float, double);
        else NF_NB(float, float);
#undef NF_NB
        NF_CHECK_LAUNCH("k_noise_blocks / k_noise_windows");
        static const bool paintRows = env_flag_default_on("NF_NOISE_PAINT_ROWS");
        if (paintRows && ws.maxBx <= NF_PR_MAXBX) {
            dim3 gr((NF_BLK + NF_PR_ROWS - 1) / NF_PR_ROWS, ws.maxBy - 1, nb);
            if (wtype == NF_F64)
                hipLaunchKernelGGL(k_noise_paint_rows<double>, gr, dim3(256), 0, s, (const double*)weights, rms, W, g, ws, stats, nbArgs);
            else
                hipLaunchKernelGGL(k_noise_paint_rows<float>, gr, dim3(256), 0, s, (const float*)weights, rms, W, g, ws, stats, nbArgs);
        } else if (wtype == NF_F64)
            hipLaunchKernelGGL(k_noise_paint_blocks<double>, gb, dim3(NF_BLK_THREADS), 0, s, (const double*)weights, rms, W, g, ws, stats, nbArgs);
        else
            hipLaunchKernelGGL(k_noise_paint_blocks<float>, gb, dim3(NF_BLK_THREADS), 0, s, (const float*)weights, rms, W, g, ws, stats, nbArgs);
        NF_CHECK_LAUNCH("k_noise_paint_blocks / k_noise_paint_rows");
        return NF_OK;
    }
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

// Map preprocessing kernels (maps.MapDict.preprocess, nemo/maps.py:178-480; maps.makeModelImage, :1658-1853).
//
//  k_median_filter    K19  ndimage.median_filter(data, size) (nemo/maps.py:366-368: the 20 x 20 moving median that is the
//                          hole-filling background): window [i - size//2, i + size - size//2 - 1] on both axes, 'reflect'
//                          borders, element of rank size*size//2 of the window.  Pure selection: bit-exact.
//                          One wavefront per 64 output pixels of a row: the vertical window of every column involved is
//                          sorted once into LDS (shared by the `size` horizontal windows it belongs to), then each thread
//                          selects the rank among its `size` sorted columns by pivoting (restricted binary searches).
//  k_disc_*                makeDegreesDistanceMap + the masking / filling statements around it (nemo/maps.py:419-424,
//                          459-465, 2479-2536): distances on the locally flat grid, holes cut into both masks, data replaced
//                          by the median-filtered background.
//  k_paint_catalog         makeModelImage's painting loop (nemo/maps.py:1790-1840 -> signals._paintSignalMap :675-688): every
//                          object's radial template inside its own window, accumulated in float32 in catalogue order (a
//                          thread owns a pixel and walks the objects in order: deterministic, one launch).
#include "nf_common.h"
#include "nf_kernels.h"

namespace nf {

__device__ __forceinline__ int pp_reflect(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * n;
    i = i % period;
    if (i < 0) i += period;
    return (i < n) ? i : period - 1 - i;
}

#define NF_MED_THREADS 64
#define NF_MED_MAXSIZE 48

template <typename T>
__global__ void __launch_bounds__(NF_MED_THREADS) k_median_filter(const T* __restrict__ in, T* __restrict__ out, int H, int W, int S) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    const int NC = NF_MED_THREADS + S - 1;
    T* srt = (T*)nf_smem;                                            // [S][NC]: sorted vertical windows, one per column
    unsigned char* st = (unsigned char*)(srt + (size_t)S * NC);      // [4][S][64]: a, b, lower / upper bound of the round
    const int t = threadIdx.x;
    const int y = blockIdx.y, x0 = blockIdx.x * NF_MED_THREADS;
    const int lo = S / 2;
    for (int c = t; c < NC; c += NF_MED_THREADS) {
        const int gx = pp_reflect(x0 - lo + c, W);
        for (int j = 0; j < S; ++j) {
            const T v = in[(long long)pp_reflect(y - lo + j, H) * W + gx];
            int i = j - 1;
            while (i >= 0 && srt[i * NC + c] > v) {
                srt[(i + 1) * NC + c] = srt[i * NC + c];
                --i;
            }
            srt[(i + 1) * NC + c] = v;
        }
    }
    __syncthreads();
    if (x0 + t >= W) return;
    unsigned char* A = st + t;                       // A[ci * 64]
    unsigned char* B = st + (size_t)S * 64 + t;
    unsigned char* LB = st + (size_t)2 * S * 64 + t;
    unsigned char* UB = st + (size_t)3 * S * 64 + t;
    for (int ci = 0; ci < S; ++ci) {
        A[ci * 64] = 0;
        B[ci * 64] = (unsigned char)S;
    }
    const int k = (S * S) / 2;
    T answer = srt[t];
    for (int round = 0; round < S * S + 8; ++round) {
        // pivot: middle of the widest remaining range
        int best = -1, bw = 0;
        for (int ci = 0; ci < S; ++ci) {
            const int w = (int)B[ci * 64] - (int)A[ci * 64];
            if (w > bw) {
                bw = w;
                best = ci;
            }
        }
        if (best < 0) break;      // (cannot happen: the ranges always contain the element of rank k)
        const T p = srt[(((int)A[best * 64] + (int)B[best * 64]) >> 1) * NC + t + best];
        int L = 0, E = 0;
        for (int ci = 0; ci < S; ++ci) {
            const T* col = srt + t + ci;
            const int a = A[ci * 64], b = B[ci * 64];
            int l = a, h = b;                       // first index in [a, b) with col >= p
            while (l < h) {
                const int m = (l + h) >> 1;
                if (col[m * NC] < p) l = m + 1; else h = m;
            }
            const int lb = l;
            h = b;                                  // first index in [lb, b) with col > p
            while (l < h) {
                const int m = (l + h) >> 1;
                if (col[m * NC] <= p) l = m + 1; else h = m;
            }
            LB[ci * 64] = (unsigned char)lb;
            UB[ci * 64] = (unsigned char)l;
            L += lb;
            E += l;
        }
        if (k < L) {
            for (int ci = 0; ci < S; ++ci) B[ci * 64] = LB[ci * 64];
        } else if (k < E) {
            answer = p;
            break;
        } else {
            for (int ci = 0; ci < S; ++ci) A[ci * 64] = UB[ci * 64];
        }
    }
    out[(long long)y * W + x0 + t] = answer;
}

int median_filter(int rtype, const void* in, void* out, int H, int W, int size, hipStream_t s) {
    if (size < 1 || size > NF_MED_MAXSIZE) {
        nf_set_error("median_filter: window size must be 1..48");
        return NF_ERR_ARG;
    }
    const size_t es = rtype == NF_F64 ? 8 : 4;
    const size_t lds = (size_t)size * (NF_MED_THREADS + size - 1) * es + (size_t)4 * size * 64;
    dim3 g(nf_div_up(W, NF_MED_THREADS), H), b(NF_MED_THREADS);
    if (rtype == NF_F64)
        hipLaunchKernelGGL(k_median_filter<double>, g, b, lds, s, (const double*)in, (double*)out, H, W, size);
    else
        hipLaunchKernelGGL(k_median_filter<float>, g, b, lds, s, (const float*)in, (float*)out, H, W, size);
    NF_CHECK_LAUNCH("k_median_filter");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// One hole: rec = {x0, y0, xPixScale, yPixScale, limit}, box = {minX, maxX, minY, maxY} as makeDegreesDistanceMap forms them.
// times60 != 0: the distance is compared in arcmin (rArcminMap = rDegMap * 60 < maskRadiusArcmin, nemo/maps.py:462-465),
// else in degrees (rDegMap < maskRadiusArcmin / 60, :422-424).  rmap (optional): the persistent distance map of
// maskPointSourcesFromCatalog (:391, never reset between rows) - k_disc_fill writes box k, k_disc_apply then tests the boxes
// 0..k against row k's limit, reading the map.
__global__ void k_disc_fill(double* __restrict__ rmap, int W, const double* __restrict__ rec, const int* __restrict__ box) {
    const double x0 = rec[0], y0 = rec[1], xs = rec[2], ys = rec[3];
    const int minX = box[0], maxX = box[1], minY = box[2], maxY = box[3];
    const int bw = maxX - minX, n = bw * (maxY - minY);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = minY + i / bw, x = minX + i % bw;
        const double xd = __dmul_rn(__dsub_rn((double)x, x0), xs), yd = __dmul_rn(__dsub_rn((double)y, y0), ys);
        rmap[(long long)y * W + x] = __dsqrt_rn(__dadd_rn(__dmul_rn(yd, yd), __dmul_rn(xd, xd)));
    }
}

template <typename T>
__global__ void k_disc_apply(T* __restrict__ data, const T* __restrict__ bck, unsigned char* __restrict__ surveyMask,
                             unsigned char* __restrict__ psMask, int W, const double* __restrict__ recs, const int* __restrict__ boxes,
                             double limit, int times60, const double* __restrict__ rmap) {
    const double* rec = recs + (size_t)blockIdx.y * 5;
    const int* box = boxes + (size_t)blockIdx.y * 4;
    const double x0 = rec[0], y0 = rec[1], xs = rec[2], ys = rec[3];
    const double lim = rmap != nullptr ? limit : rec[4];
    const int minX = box[0], maxX = box[1], minY = box[2], maxY = box[3];
    const int bw = maxX - minX, n = bw * (maxY - minY);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = minY + i / bw, x = minX + i % bw;
        const long long p = (long long)y * W + x;
        double d;
        if (rmap != nullptr) {
            d = rmap[p];
        } else {
            const double xd = __dmul_rn(__dsub_rn((double)x, x0), xs), yd = __dmul_rn(__dsub_rn((double)y, y0), ys);
            d = __dsqrt_rn(__dadd_rn(__dmul_rn(yd, yd), __dmul_rn(xd, xd)));
        }
        if (times60) d = __dmul_rn(d, 60.0);
        if (d < lim) {
            surveyMask[p] = 0;
            psMask[p] = 0;
            data[p] = bck[p];
        }
    }
}

int disc_fill(double* rmap, int W, const double* rec, const int* box, int boxPixels, hipStream_t s) {
    if (boxPixels < 1) return NF_OK;
    hipLaunchKernelGGL(k_disc_fill, dim3(nf_div_up(boxPixels, 256) > 64 ? 64 : nf_div_up(boxPixels, 256)), dim3(256), 0, s, rmap, W, rec, box);
    NF_CHECK_LAUNCH("k_disc_fill");
    return NF_OK;
}

int disc_apply(int rtype, void* data, const void* bck, unsigned char* surveyMask, unsigned char* psMask, int W, const double* recs,
               const int* boxes, int nBoxes, double limit, int times60, const double* rmap, hipStream_t s) {
    // gridDim.y holds at most 65535 boxes: a longer catalogue goes out in pieces, in catalogue order on the stream
    for (int b0 = 0; b0 < nBoxes; b0 += 65535) {
        const int nb = nBoxes - b0 < 65535 ? nBoxes - b0 : 65535;
        dim3 g(16, nb), b(256);
        const double* rc = recs + (size_t)b0 * 5;
        const int* bx = boxes + (size_t)b0 * 4;
        if (rtype == NF_F64)
            hipLaunchKernelGGL(k_disc_apply<double>, g, b, 0, s, (double*)data, (const double*)bck, surveyMask, psMask, W, rc, bx, limit,
                               times60, rmap);
        else
            hipLaunchKernelGGL(k_disc_apply<float>, g, b, 0, s, (float*)data, (const float*)bck, surveyMask, psMask, W, rc, bx, limit,
                               times60, rmap);
        NF_CHECK_LAUNCH("k_disc_apply");
    }
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// theta(hav) and the table interpolation exactly as k_paint_radial has them (nf_pointwise.hip)
__device__ __forceinline__ double pc_theta_from_hav(double h) {
    double s = __dsqrt_rn(h);
    double poly = 945.0 / 42240.0;
    poly = __dadd_rn(__dmul_rn(poly, h), 105.0 / 3456.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 15.0 / 336.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 3.0 / 40.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 1.0 / 6.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 1.0);
    return __dmul_rn(__dmul_rn(2.0, s), poly);
}

// obj[k] (8 doubles): amp, sign, f1, f2, profile id, nf, -, -.  When nf > 0 the float32 value is converted as
// maps.convertToDeltaT converts the painted clip (nemo/signals.py:684-688; numpy >= 2 promotes float32 array * float64 scalar to
// float64): (double(v) * f1) * f2, added to the float32 map in float64 and rounded back to float32.
// win[k] (8 ints): x0, x1, y0, y1, offset of A/B (per row of the window) and of C (per column) in `geom`, -, -
// prof[id]: r[ntab], prof[ntab], slope[ntab] consecutive (3 * ntab doubles per profile)
#define NF_PC_CHUNK 128
__global__ void __launch_bounds__(256) k_paint_catalog(float* __restrict__ out, int H, int W, int nObj, const double* __restrict__ obj,
                                                       const int* __restrict__ win, const double* __restrict__ geom,
                                                       const double* __restrict__ prof, int ntab, int accumulate) {
    __shared__ int swin[NF_PC_CHUNK * 4];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const bool live = x < W && y < H;
    float acc = (live && accumulate) ? out[(long long)y * W + x] : 0.0f;
    for (int k0 = 0; k0 < nObj; k0 += NF_PC_CHUNK) {
        const int nk = min(NF_PC_CHUNK, nObj - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < nk * 4; i += blockDim.x) swin[i] = win[(size_t)(k0 + i / 4) * 8 + (i & 3)];
        __syncthreads();
        if (!live) continue;
        for (int kk = 0; kk < nk; ++kk) {
            const int wx0 = swin[kk * 4], wx1 = swin[kk * 4 + 1], wy0 = swin[kk * 4 + 2], wy1 = swin[kk * 4 + 3];
            if (x < wx0 || x >= wx1 || y < wy0 || y >= wy1) continue;
            const int k = k0 + kk;
            const int* wk = win + (size_t)k * 8;
            const double* ok = obj + (size_t)k * 8;
            const double* ab = geom + wk[4];
            const int nrow = wy1 - wy0;
            const double h = __dadd_rn(ab[y - wy0], __dmul_rn(ab[nrow + (y - wy0)], geom[wk[5] + (x - wx0)]));
            const double th = pc_theta_from_hav(h);
            const double* r = prof + (size_t)((int)ok[4]) * 3 * ntab;
            const double* pf = r + ntab;
            const double* sl = pf + ntab;
            double val;
            if (th < r[0]) {
                val = pf[0];
            } else if (th > r[ntab - 1]) {
                val = 0.0;
            } else if (th == r[ntab - 1]) {
                val = pf[ntab - 1];
            } else {
                // log-uniform table: first guess from the logarithm, then exact
                const double dl = log(r[ntab - 1] / r[0]) / (double)(ntab - 1);
                int j = (int)(log(th / r[0]) / dl);
                if (j < 0) j = 0;
                if (j > ntab - 2) j = ntab - 2;
                while (j > 0 && r[j] > th) --j;
                while (j < ntab - 2 && r[j + 1] <= th) ++j;
                val = __dadd_rn(__dmul_rn(sl[j], __dsub_rn(th, r[j])), pf[j]);
            }
            float v = (float)__dmul_rn(ok[0], val);
            if (ok[1] < 0.0) v = v * -1.0f;
            if (ok[5] > 0.0)
                acc = (float)__dadd_rn((double)acc, __dmul_rn(__dmul_rn((double)v, ok[2]), ok[3]));
            else
                acc = acc + v;
        }
    }
    if (live) out[(long long)y * W + x] = acc;
}

int paint_catalog(float* out, int H, int W, int nObj, const double* obj, const int* win, const double* geom, const double* prof,
                  int ntab, int accumulate, hipStream_t s) {
    dim3 g(nf_div_up(W, 64), nf_div_up(H, 4)), b(256);
    hipLaunchKernelGGL(k_paint_catalog, g, b, 0, s, out, H, W, nObj, obj, win, geom, prof, ntab, accumulate);
    NF_CHECK_LAUNCH("k_paint_catalog");
    return NF_OK;
}

// flagMask = flagMask + (model > 1)   (nemo/maps.py:437), data = data - model (:435; float32 model, data dtype kept as
// numpy promotes: float64 data stay float64, float32 data stay float32)
template <typename T>
__global__ void k_subtract_model(T* __restrict__ data, const float* __restrict__ model, unsigned char* __restrict__ flagMask, long long n,
                                 float sign) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const float m = model[p];
        data[p] = sign < 0.0f ? data[p] - (T)m : data[p] + (T)m;
        if (flagMask != nullptr && m > 1.0f) flagMask[p] = (unsigned char)(flagMask[p] + 1);
    }
}

int add_model(int rtype, void* data, const float* model, unsigned char* flagMask, long long n, int subtract, hipStream_t s) {
    long long gl = (n + 255) / 256;
    dim3 g((unsigned)(gl > 4096 ? 4096 : gl)), b(256);
    if (rtype == NF_F64)
        hipLaunchKernelGGL(k_subtract_model<double>, g, b, 0, s, (double*)data, model, flagMask, n, subtract ? -1.0f : 1.0f);
    else
        hipLaunchKernelGGL(k_subtract_model<float>, g, b, 0, s, (float*)data, model, flagMask, n, subtract ? -1.0f : 1.0f);
    NF_CHECK_LAUNCH("k_subtract_model");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// MapDict.loadTile of the map, its weights and its three masks, and the statements of MapDict.preprocess that follow them
// (nemo/maps.py:83-175, 232-262), for maps that are resident in HBM - one pass instead of a dozen small ones per frequency
// map and tile: the clip, data[weights == 0] = 0, the survey mask zeroed outside the tile's own area, flagMask * surveyMask.
template <typename T, typename WT>
__global__ void __launch_bounds__(256) k_load_tile(const T* __restrict__ map, const WT* __restrict__ weights,
                                                   const unsigned char* __restrict__ survey, const unsigned char* __restrict__ ps,
                                                   const unsigned char* __restrict__ flag, long long pitch, int x0, int y0, int H, int W,
                                                   int ax0, int ax1, int ay0, int ay1, T* __restrict__ dataOut, WT* __restrict__ wOut,
                                                   unsigned char* __restrict__ surveyOut, unsigned char* __restrict__ psOut,
                                                   unsigned char* __restrict__ flagOut) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const long long src = (long long)(y0 + y) * pitch + (x0 + x), dst = (long long)y * W + x;
    const WT w = weights != nullptr ? weights[src] : (WT)1;
    T v = map[src];
    if (w == (WT)0) v = (T)0;                                   // :243
    dataOut[dst] = v;
    wOut[dst] = w;
    unsigned char sm;
    if (survey != nullptr) {
        sm = survey[src];
        if (ax0 >= 0 && (y < ay0 || y >= ay1 || x < ax0 || x >= ax1)) sm = 0;      // :159-164 (areaMaskInClipSection)
    } else {
        sm = w != (WT)0 ? 1 : 0;
    }
    surveyOut[dst] = sm;
    psOut[dst] = ps != nullptr ? ps[src] : (unsigned char)1;
    flagOut[dst] = flag != nullptr ? (unsigned char)(flag[src] * sm) : (unsigned char)0;      // :259-262
}

int load_tile(int rtype, const void* map, int wtype, const void* weights, const unsigned char* survey, const unsigned char* ps,
              const unsigned char* flag, long long pitch, int x0, int y0, int H, int W, int ax0, int ax1, int ay0, int ay1, void* dataOut,
              void* wOut, unsigned char* surveyOut, unsigned char* psOut, unsigned char* flagOut, hipStream_t s) {
    dim3 g((W + 255) / 256, H), b(256);
#define NF_LT(T, WT)                                                                                                              \
    hipLaunchKernelGGL((k_load_tile<T, WT>), g, b, 0, s, (const T*)map, (const WT*)weights, survey, ps, flag, pitch, x0, y0, H, W, ax0, \
                       ax1, ay0, ay1, (T*)dataOut, (WT*)wOut, surveyOut, psOut, flagOut)
    if (rtype == NF_F64 && wtype == NF_F64) NF_LT(double, double);
    else if (rtype == NF_F64) NF_LT(double, float);
    else if (wtype == NF_F64) NF_LT(float, double);
    else NF_LT(float, float);
#undef NF_LT
    NF_CHECK_LAUNCH("k_load_tile");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Source injection (nemo/maps.py:341-344): modelMap[weights == 0] = 0; data = data + modelMap - with numpy's promotion
// (float32 + float32 stays float32; a float64 operand makes the sum float64).  `base` is left as it is: an injection run
// paints thousands of catalogues into the same preprocessed tile.
template <typename TB, typename TM, typename TO>
__global__ void k_inject_model(const TB* __restrict__ base, const TM* __restrict__ model, const void* __restrict__ weights, int wIsF64,
                               TO* __restrict__ out, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        bool dead = false;
        if (weights != nullptr) dead = wIsF64 ? (((const double*)weights)[p] == 0.0) : (((const float*)weights)[p] == 0.0f);
        const TO m = dead ? (TO)0 : (TO)model[p];
        out[p] = (TO)base[p] + m;
    }
}

int inject_model(int baseType, const void* base, int modelType, const void* model, int wType, const void* weights, void* out, long long n,
                 hipStream_t s) {
    long long gl = (n + 255) / 256;
    dim3 g((unsigned)(gl > 8192 ? 8192 : gl)), b(256);
    const int w64 = wType == NF_F64 ? 1 : 0;
    if (baseType == NF_F32 && modelType == NF_F32)
        hipLaunchKernelGGL((k_inject_model<float, float, float>), g, b, 0, s, (const float*)base, (const float*)model, weights, w64, (float*)out, n);
    else if (baseType == NF_F32)
        hipLaunchKernelGGL((k_inject_model<float, double, double>), g, b, 0, s, (const float*)base, (const double*)model, weights, w64, (double*)out, n);
    else if (modelType == NF_F32)
        hipLaunchKernelGGL((k_inject_model<double, float, double>), g, b, 0, s, (const double*)base, (const float*)model, weights, w64, (double*)out, n);
    else
        hipLaunchKernelGGL((k_inject_model<double, double, double>), g, b, 0, s, (const double*)base, (const double*)model, weights, w64, (double*)out, n);
    NF_CHECK_LAUNCH("k_inject_model");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// MatchedFilter.reshapeFilter (nemo/filters.py:793-817): the float32 filter planes sampled by linear interpolation at
// fractional index positions yIdx[y'] / xIdx[x'] of the source grid (positions outside it take the edge value).
__global__ void k_resample_bilinear(const float* __restrict__ in, float* __restrict__ out, int H, int W, int H2, int W2,
                                    const double* __restrict__ yIdx, const double* __restrict__ xIdx) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W2) return;
    const float* __restrict__ pl = in + (size_t)blockIdx.z * (size_t)H * (size_t)W;
    double fy = fmin(fmax(yIdx[y], 0.0), (double)(H - 1)), fx = fmin(fmax(xIdx[x], 0.0), (double)(W - 1));
    int y0 = (int)floor(fy), x0 = (int)floor(fx);
    if (y0 > H - 2) y0 = H - 2 > 0 ? H - 2 : 0;
    if (x0 > W - 2) x0 = W - 2 > 0 ? W - 2 : 0;
    const int y1 = y0 + 1 < H ? y0 + 1 : y0, x1 = x0 + 1 < W ? x0 + 1 : x0;
    const double ty = fy - (double)y0, tx = fx - (double)x0;
    const double a = (double)pl[(size_t)y0 * W + x0], b = (double)pl[(size_t)y0 * W + x1];
    const double c = (double)pl[(size_t)y1 * W + x0], d = (double)pl[(size_t)y1 * W + x1];
    const double top = a + (b - a) * tx, bot = c + (d - c) * tx;
    out[((size_t)blockIdx.z * H2 + y) * (size_t)W2 + x] = (float)(top + (bot - top) * ty);
}

int resample_bilinear(const float* in, float* out, int N, int H, int W, int H2, int W2, const double* yIdx, const double* xIdx,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_resample_bilinear, dim3(nf_div_up(W2, 256), H2, N), dim3(256), 0, s, in, out, H, W, H2, W2, yIdx, xIdx);
    NF_CHECK_LAUNCH("k_resample_bilinear");
    return NF_OK;
}

}  // namespace nf

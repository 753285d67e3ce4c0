// Pointwise (per-pixel / per-Fourier-mode) kernels of the filter build and apply path.
// All are single-pass streaming kernels: grid-stride loops, coalesced accesses, no LDS.
//
//  k_cross_power   K4   nemo/filters.py:573      NP_ij = Re(F_i conj F_j)
//  k_build_filter  K6   nemo/filters.py:621-627  filt[:,k] = inv(noiseCov[:,:,k]) . (w |S[:,k]|)  -> float32
//  k_apply_filter  K7   nemo/filters.py:847      G = sum_i F_i * filt_i   (sum moved before the IFFT by linearity)
//  k_window_mul    K8   pixell enmap.apply_window (nemo/filters.py:100,644)
//  k_paint_radial  K20  stands in for pixell sim_objects (nemo/signals.py:681,691)
//  k_scale_mask         nemo/filters.py:853,714  filtered * signalNorm * psMask
//  k_finalize      K11  nemo/filters.py:717-756 / :1112-1157
#include "nf_common.h"
#include "nf_kernels.h"

#include <cstdlib>

namespace nf {

static inline int grid_for(long long n, int threads) {
    long long g = (n + threads - 1) / threads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------------
// Half-stored spectra (nf_fft2_forward_half): the plane has the full pitch W but only the columns c <= W/2 hold values;
// mode (y, c) with c > W/2 is the conjugate of the stored mode (-y, -c).  The consumers here need |F| or Re(F_i conj F_j),
// which do not change under a common conjugation, so they only redirect the read.
__device__ __forceinline__ long long half_src(long long p, int H, int W) {
    const unsigned y = (unsigned)(p / W), x = (unsigned)(p - (long long)y * W);
    if (2u * x <= (unsigned)W) return p;
    const unsigned ym = y == 0u ? 0u : (unsigned)H - y;
    return (long long)ym * W + ((unsigned)W - x);
}

// filtSym[n][y][x] = 0.5 (filt[n][y][x] + filt[n][-y][-x]) in float64 for the columns x <= W/2 (plane pitch W; the other
// columns are not written): real(ifft2(F * filt)) of the spectrum F of a real map is ifft2(F * filtSym).
__global__ void k_symmetrise_filter(const float* __restrict__ filt, double* __restrict__ sym, int H, int W) {
    const int hw = W / 2 + 1;
    const long long plane = (long long)blockIdx.z * H * W;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const int ym = y == 0 ? 0 : H - y;
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < hw; x += gridDim.x * blockDim.x) {
            const int xm = x == 0 ? 0 : W - x;
            sym[plane + (long long)y * W + x] = 0.5 * ((double)filt[plane + (long long)y * W + x] + (double)filt[plane + (long long)ym * W + xm]);
        }
    }
}

int symmetrise_filter(const float* filt, double* filtSym, int N, int H, int W, hipStream_t s) {
    dim3 g(nf_div_up(W / 2 + 1, 256), H < 4096 ? H : 4096, N);
    hipLaunchKernelGGL(k_symmetrise_filter, g, dim3(256), 0, s, filt, filtSym, H, W);
    NF_CHECK_LAUNCH("k_symmetrise_filter");
    return NF_OK;
}

template <typename T, bool HALF>
__global__ void k_cross_power(const cpx<T>* __restrict__ F, T* __restrict__ NP, int N, long long npix, int H, int W) {
    // NP layout [N][N][npix]; only j >= i computed, mirrored (the product is commutative bit for bit)
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        const long long ps = HALF ? half_src(p, H, W) : p;
        for (int i = 0; i < N; ++i) {
            cpx<T> a = F[(long long)i * npix + ps];
            for (int j = i; j < N; ++j) {
                cpx<T> b = F[(long long)j * npix + ps];
                // np.real(a * conj(b)) = a.x*b.x + a.y*b.y, evaluated as numpy does: two products, one add
                T v = a.x * b.x + a.y * b.y;
                NP[((long long)i * N + j) * npix + p] = v;
                if (j != i) NP[((long long)j * N + i) * npix + p] = v;
            }
        }
    }
}

template <typename T>
int cross_power(const void* F, void* NP, int N, long long npix, hipStream_t s, int halfH, int halfW) {
    if (halfH > 0)
        hipLaunchKernelGGL((k_cross_power<T, true>), dim3(grid_for(npix, 256)), dim3(256), 0, s, (const cpx<T>*)F, (T*)NP, N,
                           npix, halfH, halfW);
    else
        hipLaunchKernelGGL((k_cross_power<T, false>), dim3(grid_for(npix, 256)), dim3(256), 0, s, (const cpx<T>*)F, (T*)NP, N,
                           npix, 0, 0);
    NF_CHECK_LAUNCH("k_cross_power");
    return NF_OK;
}
template int cross_power<float>(const void*, void*, int, long long, hipStream_t, int, int);
template int cross_power<double>(const void*, void*, int, long long, hipStream_t, int, int);

// ---------------------------------------------------------------------------------------------------
// N x N inverse by LU with partial pivoting in the covariance precision T (np.linalg.inv -> LAPACK gesv),
// then a float64 matrix-vector product with w*|S| (numpy promotes the f32 inverse to f64 in np.dot),
// stored as float32.  A singular pivot leaves the mode at zero, like the reference's try/except.
// one Fourier mode p of the filter: out[i] = (inv(cov[:, :, p]) . (w |S[:, ps]|))_i as float32 (ps: where the mode's spectrum
// value is stored)
template <typename T, typename TS, int N>
__device__ __forceinline__ void build_mode(const T* __restrict__ cov, const cpx<TS>* __restrict__ S, const nf::WeightVec& w,
                                           long long npix, long long p, long long ps, float* out) {
    double rhs[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cpx<TS> s = S[(long long)i * npix + ps];
        // np.abs(complex) = hypot; computed in the template precision TS, then promoted
        TS mag = (sizeof(TS) == 4) ? (TS)hypotf((float)s.x, (float)s.y) : (TS)hypot((double)s.x, (double)s.y);
        rhs[i] = w.v[i] * (double)mag;
    }
    if (N == 1) {
        T a = cov[p];
        if (a == (T)0) {
            out[0] = 0.0f;
        } else {
            T inv = (T)1 / a;
            out[0] = (float)((double)inv * rhs[0]);
        }
    } else {
        T A[N][N], B[N][N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                A[i][j] = cov[((long long)i * N + j) * npix + p];
                B[i][j] = (i == j) ? (T)1 : (T)0;
            }
        bool singular = false;
#pragma unroll
        for (int c = 0; c < N; ++c) {
            int piv = c;
            T best = fabs(A[c][c]);
#pragma unroll
            for (int r = c + 1; r < N; ++r) {
                T v = fabs(A[r][c]);
                if (v > best) {
                    best = v;
                    piv = r;
                }
            }
            if (best == (T)0) singular = true;
#pragma unroll
            for (int r = c + 1; r < N; ++r) {
                if (r == piv) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        T t = A[c][j];
                        A[c][j] = A[r][j];
                        A[r][j] = t;
                        t = B[c][j];
                        B[c][j] = B[r][j];
                        B[r][j] = t;
                    }
                }
            }
            T rp = (T)1 / A[c][c];
#pragma unroll
            for (int r = c + 1; r < N; ++r) {
                T f = A[r][c] * rp;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    if (j > c) A[r][j] -= f * A[c][j];
                    B[r][j] -= f * B[c][j];
                }
            }
        }
        // back substitution: X = U^-1 (L^-1 P I)
#pragma unroll
        for (int c = N - 1; c >= 0; --c) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                T v = B[c][j];
#pragma unroll
                for (int k = c + 1; k < N; ++k) v -= A[c][k] * B[k][j];
                B[c][j] = v / A[c][c];
            }
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) acc += (double)B[i][j] * rhs[j];
            out[i] = singular ? 0.0f : (float)acc;
        }
    }
}

template <typename T, typename TS, int N, bool HALF>
__global__ void k_build_filter(const T* __restrict__ cov, const cpx<TS>* __restrict__ S, nf::WeightVec w,
                               float* __restrict__ filt, long long npix, int H, int W) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        float out[N];
        build_mode<T, TS, N>(cov, S, w, npix, p, HALF ? half_src(p, H, W) : p, out);
#pragma unroll
        for (int i = 0; i < N; ++i) filt[(long long)i * npix + p] = out[i];
    }
}

// The filter and its symmetrised form in one pass (half-stored spectra): a thread owns the mode (y, x <= W/2) and its mirror
// (-y, -x), computes both filter values exactly as k_build_filter does, stores them, and stores 0.5 (a + b) as
// k_symmetrise_filter would - the filter cube is not read back.  In the self-mirrored columns (x = 0, and x = W/2 for even W)
// the mirror mode belongs to another thread of the same column: its value is computed here too, but stored there.
// blockIdx.z: one filter per entry of a batch (the scales of a tile: same covariance and weights, another signal spectrum) -
// signal spectra and symmetrised filters lie batch entry after batch entry, the float32 cubes are the callers' (fp.p[z], or null).
template <typename T, typename TS, int N>
__global__ void k_build_filter_sym(const T* __restrict__ cov, const cpx<TS>* __restrict__ S0, nf::WeightVec w,
                                   nf::BatchPtrs fp, double* __restrict__ sym0, int H, int W) {
    const int hw = W / 2 + 1;
    const long long npix = (long long)H * W;
    const cpx<TS>* __restrict__ S = S0 + (long long)blockIdx.z * N * npix;
    double* __restrict__ sym = sym0 + (long long)blockIdx.z * N * npix;
    float* __restrict__ filt = (float*)fp.p[blockIdx.z];
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const int ym = y == 0 ? 0 : H - y;
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < hw; x += gridDim.x * blockDim.x) {
            const int xm = x == 0 ? 0 : W - x;
            const long long p = (long long)y * W + x, pm = (long long)ym * W + xm;
            float o1[N], o2[N];
            build_mode<T, TS, N>(cov, S, w, npix, p, p, o1);
            const bool selfColumn = xm == x;                  // the mirror lies in this column (stored as it is)
            if (pm == p) {
#pragma unroll
                for (int i = 0; i < N; ++i) o2[i] = o1[i];
            } else {
                build_mode<T, TS, N>(cov, S, w, npix, pm, selfColumn ? pm : p, o2);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (filt != nullptr) {
                    filt[(long long)i * npix + p] = o1[i];
                    if (!selfColumn) filt[(long long)i * npix + pm] = o2[i];
                }
                sym[(long long)i * npix + p] = 0.5 * ((double)o1[i] + (double)o2[i]);
            }
        }
    }
}

// The scales of a tile share the noise covariance and the weights: the N x N inverse of a mode is the same for every batch
// entry, only the signal spectrum differs.  k_build_filter_sym (blockIdx.z = entry) reads and inverts the covariance of a mode
// and of its mirror once per entry; here a thread inverts both once and walks the entries (N <= 3: the two inverses stay in
// registers).  build_mode's statements, split at the point where the right-hand side enters: the same bits.
template <typename T, int N>
__device__ __forceinline__ void invert_mode(const T* __restrict__ cov, long long npix, long long p, T (&B)[N][N], bool& singular) {
    singular = false;
    if (N == 1) {
        const T a = cov[p];
        singular = a == (T)0;
        B[0][0] = singular ? (T)0 : (T)1 / a;
        return;
    }
    T A[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            A[i][j] = cov[((long long)i * N + j) * npix + p];
            B[i][j] = (i == j) ? (T)1 : (T)0;
        }
#pragma unroll
    for (int c = 0; c < N; ++c) {
        int piv = c;
        T best = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            T v = fabs(A[r][c]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        if (best == (T)0) singular = true;
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    T t = A[c][j];
                    A[c][j] = A[r][j];
                    A[r][j] = t;
                    t = B[c][j];
                    B[c][j] = B[r][j];
                    B[r][j] = t;
                }
            }
        }
        T rp = (T)1 / A[c][c];
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            T f = A[r][c] * rp;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (j > c) A[r][j] -= f * A[c][j];
                B[r][j] -= f * B[c][j];
            }
        }
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            T v = B[c][j];
#pragma unroll
            for (int k = c + 1; k < N; ++k) v -= A[c][k] * B[k][j];
            B[c][j] = v / A[c][c];
        }
    }
}

template <typename T, typename TS, int N>
__device__ __forceinline__ void apply_mode(const T (&B)[N][N], bool singular, const cpx<TS>* __restrict__ S, const nf::WeightVec& w,
                                           long long npix, long long ps, float* out) {
    double rhs[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        cpx<TS> s = S[(long long)i * npix + ps];
        TS mag = (sizeof(TS) == 4) ? (TS)hypotf((float)s.x, (float)s.y) : (TS)hypot((double)s.x, (double)s.y);
        rhs[i] = w.v[i] * (double)mag;
    }
    if (N == 1) {
        out[0] = singular ? 0.0f : (float)((double)B[0][0] * rhs[0]);
        return;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) acc += (double)B[i][j] * rhs[j];
        out[i] = singular ? 0.0f : (float)acc;
    }
}

template <typename T, typename TS, int N>
__global__ void k_build_filter_sym_shared(const T* __restrict__ cov, const cpx<TS>* __restrict__ S0, nf::WeightVec w,
                                          nf::BatchPtrs fp, double* __restrict__ sym0, int H, int W, int nb) {
    const int hw = W / 2 + 1;
    const long long npix = (long long)H * W;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const int ym = y == 0 ? 0 : H - y;
        for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < hw; x += gridDim.x * blockDim.x) {
            const int xm = x == 0 ? 0 : W - x;
            const long long p = (long long)y * W + x, pm = (long long)ym * W + xm;
            const bool selfColumn = xm == x;                  // the mirror lies in this column (stored as it is)
            T B1[N][N], B2[N][N];
            bool s1, s2 = false;
            invert_mode<T, N>(cov, npix, p, B1, s1);
            if (pm != p) invert_mode<T, N>(cov, npix, pm, B2, s2);
            for (int z = 0; z < nb; ++z) {
                const cpx<TS>* __restrict__ S = S0 + (long long)z * N * npix;
                double* __restrict__ sym = sym0 + (long long)z * N * npix;
                float* __restrict__ filt = (float*)fp.p[z];
                float o1[N], o2[N];
                apply_mode<T, TS, N>(B1, s1, S, w, npix, p, o1);
                if (pm == p) {
#pragma unroll
                    for (int i = 0; i < N; ++i) o2[i] = o1[i];
                } else {
                    apply_mode<T, TS, N>(B2, s2, S, w, npix, selfColumn ? pm : p, o2);
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    if (filt != nullptr) {
                        filt[(long long)i * npix + p] = o1[i];
                        if (!selfColumn) filt[(long long)i * npix + pm] = o2[i];
                    }
                    sym[(long long)i * npix + p] = 0.5 * ((double)o1[i] + (double)o2[i]);
                }
            }
        }
    }
}

template <typename T, typename TS>
static int build_filter_n(const void* cov, const void* S, const WeightVec& w, float* filt, int N, long long npix, hipStream_t s,
                          int halfH, int halfW) {
    dim3 g(grid_for(npix, 256)), b(256);
#define NF_BF(NN)                                                                                                       \
    case NN:                                                                                                            \
        if (halfH > 0)                                                                                                  \
            hipLaunchKernelGGL((k_build_filter<T, TS, NN, true>), g, b, 0, s, (const T*)cov, (const cpx<TS>*)S, w, filt, npix, \
                               halfH, halfW);                                                                           \
        else                                                                                                            \
            hipLaunchKernelGGL((k_build_filter<T, TS, NN, false>), g, b, 0, s, (const T*)cov, (const cpx<TS>*)S, w, filt, npix, \
                               0, 0);                                                                                   \
        break;
    switch (N) {
        NF_BF(1)
        NF_BF(2)
        NF_BF(3)
        NF_BF(4)
        NF_BF(5)
        NF_BF(6)
        default:
            nf_set_error("build_filter supports 1..6 frequencies");
            return NF_ERR_ARG;
    }
#undef NF_BF
    NF_CHECK_LAUNCH("k_build_filter");
    return NF_OK;
}

// NF_BUILD_SHARED=0: one inverse per entry (k_build_filter_sym) always
bool build_filter_shares_inverse(int nb, int N) {
    static const int shared = getenv("NF_BUILD_SHARED") ? atoi(getenv("NF_BUILD_SHARED")) : 1;
    return shared && nb > 1 && N <= 3;
}

template <typename T, typename TS>
static int build_filter_sym_n(const void* cov, const void* S, const WeightVec& w, const BatchPtrs& fp, int nb, double* sym, int N, int H,
                              int W, hipStream_t s) {
    if (build_filter_shares_inverse(nb, N)) {
        // one inverse per mode for all entries (64-thread workgroups: the grid has no z to fill the chip with)
        dim3 gs(nf_div_up(W / 2 + 1, 64), H < 4096 ? H : 4096, 1), bs(64);
#define NF_BFSH(NN)                                                                                                            \
    case NN:                                                                                                                   \
        hipLaunchKernelGGL((k_build_filter_sym_shared<T, TS, NN>), gs, bs, 0, s, (const T*)cov, (const cpx<TS>*)S, w, fp, sym, H, W, nb); \
        break;
        switch (N) {
            NF_BFSH(1)
            NF_BFSH(2)
            NF_BFSH(3)
            default: break;
        }
#undef NF_BFSH
        NF_CHECK_LAUNCH("k_build_filter_sym_shared");
        return NF_OK;
    }
    dim3 g(nf_div_up(W / 2 + 1, 256), H < 4096 ? H : 4096, nb), b(256);
#define NF_BFS(NN)                                                                                                             \
    case NN:                                                                                                                   \
        hipLaunchKernelGGL((k_build_filter_sym<T, TS, NN>), g, b, 0, s, (const T*)cov, (const cpx<TS>*)S, w, fp, sym, H, W);   \
        break;
    switch (N) {
        NF_BFS(1)
        NF_BFS(2)
        NF_BFS(3)
        NF_BFS(4)
        NF_BFS(5)
        NF_BFS(6)
        default:
            nf_set_error("build_filter supports 1..6 frequencies");
            return NF_ERR_ARG;
    }
#undef NF_BFS
    NF_CHECK_LAUNCH("k_build_filter_sym");
    return NF_OK;
}

int build_filter_sym_batch(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, const BatchPtrs& filt, int nb,
                           double* sym, int N, int H, int W, hipStream_t s) {
    if (nb < 1 || nb > NF_SB_MAX) {
        nf_set_error("build_filter_sym_batch: 1..8 filters per launch");
        return NF_ERR_ARG;
    }
    if (covType == NF_F64 && sigType == NF_F32) return build_filter_sym_n<double, float>(cov, S, w, filt, nb, sym, N, H, W, s);
    if (covType == NF_F64 && sigType == NF_F64) return build_filter_sym_n<double, double>(cov, S, w, filt, nb, sym, N, H, W, s);
    if (covType == NF_F32 && sigType == NF_F32) return build_filter_sym_n<float, float>(cov, S, w, filt, nb, sym, N, H, W, s);
    if (covType == NF_F32 && sigType == NF_F64) return build_filter_sym_n<float, double>(cov, S, w, filt, nb, sym, N, H, W, s);
    nf_set_error("build_filter: bad dtype");
    return NF_ERR_ARG;
}

int build_filter_sym(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, float* filt, double* sym, int N,
                     int H, int W, hipStream_t s) {
    BatchPtrs fp;
    memset(&fp, 0, sizeof(fp));
    fp.p[0] = filt;
    return build_filter_sym_batch(covType, sigType, cov, S, w, fp, 1, sym, N, H, W, s);
}

int build_filter(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, float* filt, int N,
                 long long npix, hipStream_t s, int halfH, int halfW) {
    if (covType == NF_F64 && sigType == NF_F32) return build_filter_n<double, float>(cov, S, w, filt, N, npix, s, halfH, halfW);
    if (covType == NF_F64 && sigType == NF_F64) return build_filter_n<double, double>(cov, S, w, filt, N, npix, s, halfH, halfW);
    if (covType == NF_F32 && sigType == NF_F32) return build_filter_n<float, float>(cov, S, w, filt, N, npix, s, halfH, halfW);
    if (covType == NF_F32 && sigType == NF_F64) return build_filter_n<float, double>(cov, S, w, filt, N, npix, s, halfH, halfW);
    nf_set_error("build_filter: bad dtype");
    return NF_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_apply_filter(const cpx<T>* __restrict__ F, const float* __restrict__ filt, cpx<T>* __restrict__ G, int N,
                               int sum, long long npix) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
        if (sum) {
            T re = 0, im = 0;
            for (int i = 0; i < N; ++i) {
                cpx<T> f = F[(long long)i * npix + p];
                T h = (T)filt[(long long)i * npix + p];
                re += f.x * h;
                im += f.y * h;
            }
            cpx<T> g;
            g.x = re;
            g.y = im;
            G[p] = g;
        } else {
            for (int i = 0; i < N; ++i) {
                cpx<T> f = F[(long long)i * npix + p];
                T h = (T)filt[(long long)i * npix + p];
                cpx<T> g;
                g.x = f.x * h;
                g.y = f.y * h;
                G[(long long)i * npix + p] = g;
            }
        }
    }
}

template <typename T>
int apply_filter(const void* F, const float* filt, void* G, int N, int sum, long long npix, hipStream_t s) {
    hipLaunchKernelGGL(k_apply_filter<T>, dim3(grid_for(npix, 256)), dim3(256), 0, s, (const cpx<T>*)F, filt, (cpx<T>*)G, N,
                       sum, npix);
    NF_CHECK_LAUNCH("k_apply_filter");
    return NF_OK;
}
template int apply_filter<float>(const void*, const float*, void*, int, int, long long, hipStream_t);
template int apply_filter<double>(const void*, const float*, void*, int, int, long long, hipStream_t);

// ---------------------------------------------------------------------------------------------------
// F[b][y][x] *= wy[y] * wx[x]   (window tables already raised to the requested power on the host, as
// enmap.apply_window does: wy[:,None]**pow * wx[None,:]**pow, product formed first, in float64)
template <typename T>
__global__ void k_window_mul(cpx<T>* __restrict__ F, const double* __restrict__ wy, const double* __restrict__ wx, int H,
                             int W, int batch) {
    long long n = (long long)H * W * batch;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        long long q = p % ((long long)H * W);
        int y = (int)(q / W), x = (int)(q % W);
        double win = wy[y] * wx[x];
        cpx<T> f = F[p];
        // numpy in-place complex *= float64: computed in complex128, cast back to T
        f.x = (T)((double)f.x * win);
        f.y = (T)((double)f.y * win);
        F[p] = f;
    }
}

template <typename T>
int window_mul(void* F, const double* wy, const double* wx, int H, int W, int batch, hipStream_t s) {
    hipLaunchKernelGGL(k_window_mul<T>, dim3(grid_for((long long)H * W * batch, 256)), dim3(256), 0, s, (cpx<T>*)F, wy, wx, H,
                       W, batch);
    NF_CHECK_LAUNCH("k_window_mul");
    return NF_OK;
}
template int window_mul<float>(void*, const double*, const double*, int, int, int, hipStream_t);
template int window_mul<double>(void*, const double*, const double*, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------
// Radial template painter.  hav = A[y] + B[y]*C[x]; theta = 2 asin(sqrt(hav)) by its series to h^5 (only
// + * sqrt, with explicitly unfused roundings so the result is bit-identical to the numpy statement);
// linear interpolation on the log-uniform table r[], prof[].
__device__ __forceinline__ double theta_from_hav(double h) {
    double s = __dsqrt_rn(h);
    double poly = 945.0 / 42240.0;
    poly = __dadd_rn(__dmul_rn(poly, h), 105.0 / 3456.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 15.0 / 336.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 3.0 / 40.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 1.0 / 6.0);
    poly = __dadd_rn(__dmul_rn(poly, h), 1.0);
    return __dmul_rn(__dmul_rn(2.0, s), poly);
}

// DUAL: a second template from a second profile / slope table on the same radii, painted in the same pass (the matched
// filter paints the unit template and the amplitude-scaled one at the same place: the haversine, its arcsine and the
// table search are shared, only the two interpolations differ).
template <bool DUAL>
__global__ void __launch_bounds__(256) k_paint_radial(float* __restrict__ out, float* __restrict__ out2, int H, int W,
                                                      const double* __restrict__ A, const double* __restrict__ B,
                                                      const double* __restrict__ C, const double* __restrict__ r,
                                                      const double* __restrict__ prof, const double* __restrict__ slope,
                                                      const double* __restrict__ prof2, const double* __restrict__ slope2, int ntab,
                                                      float log2r0, float invDlog2, double amp, float sign, double amp2,
                                                      float sign2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const double rlast = r[ntab - 1];
    const double rfirst = r[0];
    const double cx = C[x];
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        double h = __dadd_rn(A[y], __dmul_rn(B[y], cx));
        double th = theta_from_hav(h);
        double val, val2 = 0.0;
        if (th < rfirst) {
            val = prof[0];
            if (DUAL) val2 = prof2[0];
        } else if (th > rlast) {
            val = 0.0;
        } else if (th == rlast) {
            val = prof[ntab - 1];
            if (DUAL) val2 = prof2[ntab - 1];
        } else {
            // interval index from a float32 log2 (the table is log-uniform); the two loops below make it exact
            int j = (int)((__log2f((float)th) - log2r0) * invDlog2);
            if (j < 0) j = 0;
            if (j > ntab - 2) j = ntab - 2;
            while (j > 0 && r[j] > th) --j;
            while (j < ntab - 2 && r[j + 1] <= th) ++j;
            const double dr = __dsub_rn(th, r[j]);
            const double sl = slope != nullptr ? slope[j] : __ddiv_rn(__dsub_rn(prof[j + 1], prof[j]), __dsub_rn(r[j + 1], r[j]));
            val = __dadd_rn(__dmul_rn(sl, dr), prof[j]);
            if (DUAL) {
                const double sl2 = slope2 != nullptr ? slope2[j] : __ddiv_rn(__dsub_rn(prof2[j + 1], prof2[j]), __dsub_rn(r[j + 1], r[j]));
                val2 = __dadd_rn(__dmul_rn(sl2, dr), prof2[j]);
            }
        }
        float v = (float)__dmul_rn(amp, val);
        out[(long long)y * W + x] = (sign < 0.0f) ? v * -1.0f : v;
        if (DUAL) {
            float v2 = (float)__dmul_rn(amp2, val2);
            out2[(long long)y * W + x] = (sign2 < 0.0f) ? v2 * -1.0f : v2;
        }
    }
}

int paint_radial(float* out, int H, int W, const double* A, const double* B, const double* C, const double* r,
                 const double* prof, const double* slope, int ntab, double r0, double dlnr, double amp, double sign,
                 hipStream_t s) {
    const int gy = H < 1024 ? H : 1024;
    hipLaunchKernelGGL(k_paint_radial<false>, dim3((W + 255) / 256, gy), dim3(256), 0, s, out, (float*)nullptr, H, W, A, B, C, r, prof,
                       slope, (const double*)nullptr, (const double*)nullptr, ntab, (float)(log(r0) / log(2.0)),
                       (float)(1.0 / (dlnr / log(2.0))), amp, (float)sign, 0.0, 1.0f);
    NF_CHECK_LAUNCH("k_paint_radial");
    return NF_OK;
}

// The templates of several frequency maps of one tile in one launch: the angular distance of a pixel (the costly part) is the
// same for all of them, only the radial tables differ.  Plane i goes to out + i * H * W; per plane exactly the arithmetic of
// k_paint_radial<false>.
__global__ void __launch_bounds__(256) k_paint_radial_planes(float* __restrict__ out, int H, int W, const double* __restrict__ A,
                                                             const double* __restrict__ B, const double* __restrict__ C,
                                                             PaintPlanes P) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const double cx = C[x];
    const long long npix = (long long)H * W;
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const double h = __dadd_rn(A[y], __dmul_rn(B[y], cx));
        const double th = theta_from_hav(h);
        const float lth = __log2f((float)th);
        // planes on the same radii (the templates of a bank: one log-uniform grid) share the interval search: cls = where th
        // lies (0 below the table, 1 beyond it, 2 on its last radius, 3 inside: interval j, offset dr)
        const double* rSeen = nullptr;
        int cls = 0, j = 0;
        double dr = 0.0;
        for (int i = 0; i < P.n; ++i) {
            const double* __restrict__ r = P.r[i];
            const double* __restrict__ prof = P.prof[i];
            const double* __restrict__ slope = P.slope[i];
            const int ntab = P.ntab[i];
            if (r != rSeen) {
                rSeen = r;
                const double rlast = r[ntab - 1], rfirst = r[0];
                if (th < rfirst) {
                    cls = 0;
                } else if (th > rlast) {
                    cls = 1;
                } else if (th == rlast) {
                    cls = 2;
                } else {
                    cls = 3;
                    j = (int)((lth - P.log2r0[i]) * P.invDlog2[i]);
                    if (j < 0) j = 0;
                    if (j > ntab - 2) j = ntab - 2;
                    while (j > 0 && r[j] > th) --j;
                    while (j < ntab - 2 && r[j + 1] <= th) ++j;
                    dr = __dsub_rn(th, r[j]);
                }
            }
            double val;
            if (cls == 0) {
                val = prof[0];
            } else if (cls == 1) {
                val = 0.0;
            } else if (cls == 2) {
                val = prof[ntab - 1];
            } else {
                const double sl = slope != nullptr ? slope[j] : __ddiv_rn(__dsub_rn(prof[j + 1], prof[j]), __dsub_rn(r[j + 1], r[j]));
                val = __dadd_rn(__dmul_rn(sl, dr), prof[j]);
            }
            const float v = (float)__dmul_rn(P.amp[i], val);
            out[(long long)i * npix + (long long)y * W + x] = (P.sign[i] < 0.0f) ? v * -1.0f : v;
        }
    }
}

int paint_radial_planes(float* out, int H, int W, const double* A, const double* B, const double* C, const PaintPlanes& P,
                        hipStream_t s) {
    const int gy = H < 1024 ? H : 1024;
    hipLaunchKernelGGL(k_paint_radial_planes, dim3((W + 255) / 256, gy), dim3(256), 0, s, out, H, W, A, B, C, P);
    NF_CHECK_LAUNCH("k_paint_radial_planes");
    return NF_OK;
}

int paint_radial2(float* out, float* out2, int H, int W, const double* A, const double* B, const double* C, const double* r,
                  const double* prof, const double* slope, const double* prof2, const double* slope2, int ntab, double r0,
                  double dlnr, double amp, double sign, double amp2, double sign2, hipStream_t s) {
    const int gy = H < 1024 ? H : 1024;
    hipLaunchKernelGGL(k_paint_radial<true>, dim3((W + 255) / 256, gy), dim3(256), 0, s, out, out2, H, W, A, B, C, r, prof, slope,
                       prof2, slope2, ntab, (float)(log(r0) / log(2.0)), (float)(1.0 / (dlnr / log(2.0))), amp, (float)sign, amp2,
                       (float)sign2);
    NF_CHECK_LAUNCH("k_paint_radial");
    return NF_OK;
}

// out = in[0] + in[1] + ... + in[N-1], float32 planes added in plane order (ndarray.sum(axis=0) of a [N, H, W] cube adds the
// planes one after the other): nemo/filters.py:847 sums the per-frequency filtered maps AFTER the inverse transforms.
__global__ void k_sum_planes(const float* __restrict__ in, float* __restrict__ out, int N, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        float acc = in[p];
        for (int i = 1; i < N; ++i) acc = __fadd_rn(acc, in[(long long)i * n + p]);
        out[p] = acc;
    }
}

int sum_planes(const float* in, float* out, int N, long long n, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_planes, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, N, n);
    NF_CHECK_LAUNCH("k_sum_planes");
    return NF_OK;
}

// filtered(VT) = float32 map * signalNorm * psMask.  VT = double when signalNorm is a float64 scalar
// ('yc' mode under numpy >= 2 promotion), float when it is a float32 scalar ('uK' mode).
template <typename VT>
__global__ void k_scale_mask(const float* __restrict__ in, VT* __restrict__ out, double norm, const double* __restrict__ normDev,
                             const unsigned char* __restrict__ psMask, long long n) {
    if (normDev != nullptr) norm = *normDev;      // normalisation left on the device by the caller (no host round trip)
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        VT v = (VT)in[p] * (VT)norm;
        if (psMask != nullptr) v = v * (VT)psMask[p];
        out[p] = v;
    }
}

int scale_mask(int vtype, const float* in, void* out, double norm, const double* normDev, const unsigned char* psMask,
               long long n, hipStream_t s) {
    dim3 g(grid_for(n, 256)), b(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_scale_mask<double>, g, b, 0, s, in, (double*)out, norm, normDev, psMask, n);
    else
        hipLaunchKernelGGL(k_scale_mask<float>, g, b, 0, s, in, (float*)out, norm, normDev, psMask, n);
    NF_CHECK_LAUNCH("k_scale_mask");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// nemo/pipelines.py:244-247 (cached RMS map branch): SN = float64(data); SN[RMS > 0] /= RMS
template <typename VT>
__global__ void k_sn_cached(const VT* __restrict__ data, const float* __restrict__ rms, double* __restrict__ sn, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        double v = (double)data[p];
        const float r = rms[p];
        if (r > 0.0f) v = v / (double)r;
        sn[p] = v;
    }
}

int sn_cached(int vtype, const void* data, const float* rms, double* sn, long long n, hipStream_t s) {
    dim3 g(grid_for(n, 256)), b(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_sn_cached<double>, g, b, 0, s, (const double*)data, rms, sn, n);
    else
        hipLaunchKernelGGL(k_sn_cached<float>, g, b, 0, s, (const float*)data, rms, sn, n);
    NF_CHECK_LAUNCH("k_sn_cached");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Tail of buildAndApply: S/N map, edge-check / survey / point-source / apodisation masks.
//   SN = filtered / RMS where RMS > 0 (else filtered);  edge = edgeCheck (or 1)
//   filtered' = filtered*edge [*surveyMask' if fourier variant];  surveyMask' = edge*surveyMask*psMask
//   surveyMask'' = surveyMask' * apodMask;  SN' = SN*surveyMask'' (NaN -> 0);  RMS' = RMS*surveyMask''
// blockIdx.y: one map per entry of a batch (the scales of a tile share the masks; every other operand is per map, fb.*[y]).
template <typename VT>
__global__ void __launch_bounds__(256) k_finalize(nf::FinalizeBatch fb, const unsigned char* __restrict__ surveyMask,
                           const unsigned char* __restrict__ psMask, int H, int W, int apodPix, int mulSurvey, int nchunks) {
    VT* __restrict__ filtered = (VT*)fb.filtered[blockIdx.y];
    const float* rms = fb.rms[blockIdx.y];
    const unsigned char* __restrict__ edgeCheck = fb.edgeCheck[blockIdx.y];
    VT* __restrict__ snOut = (VT*)fb.snOut[blockIdx.y];
    float* rmsOut = fb.rmsOut[blockIdx.y];
    unsigned char* __restrict__ maskOut = fb.maskOut[blockIdx.y];
    unsigned long long* __restrict__ chunkRange = fb.chunkRange[blockIdx.y];
    int* __restrict__ zeroRows = fb.zeroRows[blockIdx.y];
    const long long n = (long long)H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wavesPerWG = blockDim.x >> 6;
    // a wavefront takes one 1024-pixel chunk after another, 64 consecutive pixels at a time: the chunk's largest and smallest
    // S/N stay in registers and leave through one plain store each (no atomics, nothing to clear beforehand)
    for (long long c = (long long)blockIdx.x * wavesPerWG + wave; c < nchunks; c += (long long)gridDim.x * wavesPerWG) {
        double mx = -INFINITY, mn = INFINITY;
        // pixel coordinates by 32-bit arithmetic, advanced by 64 per step (H * W < 2^31: host check)
        const unsigned p0 = (unsigned)c * NF_SN_CHUNK + (unsigned)lane;
        int y = (int)(p0 / (unsigned)W), x = (int)(p0 - (unsigned)y * (unsigned)W);
        const int ay = min(apodPix, H), ax = min(apodPix, W);
#pragma unroll 4
        for (int it = 0; it < NF_SN_CHUNK / 64; ++it) {
            const unsigned p = p0 + (unsigned)it * 64u;
            if (it > 0) {
                x += 64;
                while (x >= W) {
                    x -= W;
                    ++y;
                }
            }
            if (p >= (unsigned)n) continue;
            const VT fIn = filtered[p];
            VT f = fIn;
            float r = rms[p];
            VT sn = f;
            if (r > 0.0f) sn = f / (VT)r;
            unsigned char e = edgeCheck ? edgeCheck[p] : (unsigned char)1;
            unsigned char sm = (unsigned char)(e * (surveyMask ? surveyMask[p] : 1) * (psMask ? psMask[p] : 1));
            f = f * (VT)e;
            if (mulSurvey) f = f * (VT)sm;
            // apod(ones, w) == 1 only at distance >= w-1 from every edge
            bool inside = (ay <= 0 || (y >= ay - 1 && y <= H - ay)) && (ax <= 0 || (x >= ax - 1 && x <= W - ax));
            unsigned char fm = inside ? sm : (unsigned char)0;
            sn = sn * (VT)fm;
            if (sn != sn) sn = (VT)0;
            // stores of values that are already there are left out (x * 1 is x, bit for bit): with masks that are mostly one the
            // filtered map - and the noise map, when it is updated in place - cost no write traffic
            if (f != fIn || f != f) filtered[p] = f;
            if (zeroRows != nullptr && f == (VT)0) zeroRows[y] = 1;     // (rows of the filtered map that hold a zero)
            snOut[p] = sn;
            const float rNew = r * (float)fm;
            if (rmsOut != rms || rNew != r || rNew != rNew) rmsOut[p] = rNew;
            maskOut[p] = fm;
            mx = fmax(mx, (double)sn);
            mn = fmin(mn, (double)sn);
        }
        if (chunkRange != nullptr) {
            for (int off = 32; off > 0; off >>= 1) {
                mx = fmax(mx, __shfl_xor(mx, off, 64));
                mn = fmin(mn, __shfl_xor(mn, off, 64));
            }
            if (lane == 0) {
                chunkRange[c] = nf_ordered_key(mx);
                chunkRange[nchunks + c] = nf_ordered_key(mn);
            }
        }
    }
}

// (the zeroRows arrays of a batch are cleared by the caller)
int finalize_batch(int vtype, const FinalizeBatch& fb, int nb, const unsigned char* surveyMask, const unsigned char* psMask, int H, int W,
                   int apodPix, int mulSurvey, hipStream_t s) {
    if ((long long)H * W >= 2147483647LL) {
        nf_set_error("finalize: map too large for 32-bit pixel indices");
        return NF_ERR_SIZE;
    }
    if (nb < 1 || nb > NF_SB_MAX) {
        nf_set_error("finalize_batch: 1..8 maps per launch");
        return NF_ERR_ARG;
    }
    const int nchunks = nf_div_up((long long)H * W, NF_SN_CHUNK);
    dim3 g(grid_for((long long)nchunks * 64, 256), nb), b(256);        // one wavefront per chunk, as many at a time as the grid holds
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_finalize<double>, g, b, 0, s, fb, surveyMask, psMask, H, W, apodPix, mulSurvey, nchunks);
    else
        hipLaunchKernelGGL(k_finalize<float>, g, b, 0, s, fb, surveyMask, psMask, H, W, apodPix, mulSurvey, nchunks);
    NF_CHECK_LAUNCH("k_finalize");
    return NF_OK;
}

int finalize(int vtype, void* filtered, const float* rms, const unsigned char* surveyMask, const unsigned char* psMask,
             const unsigned char* edgeCheck, void* snOut, float* rmsOut, unsigned char* maskOut, int H, int W, int apodPix,
             int mulSurvey, unsigned long long* chunkRange, int* zeroRows, hipStream_t s) {
    if (zeroRows != nullptr) NF_HIP(hipMemsetAsync(zeroRows, 0, (size_t)H * sizeof(int), s));
    FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));
    fb.filtered[0] = filtered;
    fb.rms[0] = rms;
    fb.edgeCheck[0] = edgeCheck;
    fb.snOut[0] = snOut;
    fb.rmsOut[0] = rmsOut;
    fb.maskOut[0] = maskOut;
    fb.chunkRange[0] = chunkRange;
    fb.zeroRows[0] = zeroRows;
    return finalize_batch(vtype, fb, 1, surveyMask, psMask, H, W, apodPix, mulSurvey, s);
}

// ---------------------------------------------------------------------------------------------------
// Tail of buildAndApply followed by the driver's cached-noise branch (nemo/filters.py:725-748 then nemo/pipelines.py:243-247), for
// the re-filters of an injection run: the masks exactly as k_finalize forms them, no noise map of this pass, and
//   SN (float64) = filtered' / cachedRMS where cachedRMS > 0, else filtered'      (filtered' = the masked filtered map)
template <typename VT>
__global__ void __launch_bounds__(256) k_finalize_cached(VT* __restrict__ filtered, const float* __restrict__ rms,
                                                         const unsigned char* __restrict__ surveyMask, const unsigned char* __restrict__ psMask,
                                                         const unsigned char* __restrict__ edgeCheck, double* __restrict__ snOut,
                                                         unsigned char* __restrict__ maskOut, int H, int W, int apodPix, int mulSurvey,
                                                         unsigned long long* __restrict__ chunkRange, int nchunks, int* __restrict__ zeroRows) {
    const long long n = (long long)H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wavesPerWG = blockDim.x >> 6;
    for (long long c = (long long)blockIdx.x * wavesPerWG + wave; c < nchunks; c += (long long)gridDim.x * wavesPerWG) {
        double mx = -INFINITY, mn = INFINITY;
        const unsigned p0 = (unsigned)c * NF_SN_CHUNK + (unsigned)lane;
        int y = (int)(p0 / (unsigned)W), x = (int)(p0 - (unsigned)y * (unsigned)W);
        const int ay = min(apodPix, H), ax = min(apodPix, W);
#pragma unroll 4
        for (int it = 0; it < NF_SN_CHUNK / 64; ++it) {
            const unsigned p = p0 + (unsigned)it * 64u;
            if (it > 0) {
                x += 64;
                while (x >= W) {
                    x -= W;
                    ++y;
                }
            }
            if (p >= (unsigned)n) continue;
            const VT fIn = filtered[p];
            VT f = fIn;
            unsigned char e = edgeCheck ? edgeCheck[p] : (unsigned char)1;
            unsigned char sm = (unsigned char)(e * (surveyMask ? surveyMask[p] : 1) * (psMask ? psMask[p] : 1));
            f = f * (VT)e;
            if (mulSurvey) f = f * (VT)sm;
            bool inside = (ay <= 0 || (y >= ay - 1 && y <= H - ay)) && (ax <= 0 || (x >= ax - 1 && x <= W - ax));
            if (f != fIn || f != f) filtered[p] = f;
            if (zeroRows != nullptr && f == (VT)0) zeroRows[y] = 1;
            double sn = (double)f;
            const float r = rms[p];
            if (r > 0.0f) sn = sn / (double)r;
            snOut[p] = sn;
            maskOut[p] = inside ? sm : (unsigned char)0;
            // (a NaN never exceeds a threshold: it stays out of the chunk's range, as fmax / fmin leave it out)
            mx = fmax(mx, sn);
            mn = fmin(mn, sn);
        }
        if (chunkRange != nullptr) {
            for (int off = 32; off > 0; off >>= 1) {
                mx = fmax(mx, __shfl_xor(mx, off, 64));
                mn = fmin(mn, __shfl_xor(mn, off, 64));
            }
            if (lane == 0) {
                chunkRange[c] = nf_ordered_key(mx);
                chunkRange[nchunks + c] = nf_ordered_key(mn);
            }
        }
    }
}

int finalize_cached(int vtype, void* filtered, const float* rms, const unsigned char* surveyMask, const unsigned char* psMask,
                    const unsigned char* edgeCheck, double* snOut, unsigned char* maskOut, int H, int W, int apodPix, int mulSurvey,
                    unsigned long long* chunkRange, int* zeroRows, hipStream_t s) {
    if ((long long)H * W >= 2147483647LL) {
        nf_set_error("finalize_cached: map too large for 32-bit pixel indices");
        return NF_ERR_SIZE;
    }
    const int nchunks = nf_div_up((long long)H * W, NF_SN_CHUNK);
    if (zeroRows != nullptr) NF_HIP(hipMemsetAsync(zeroRows, 0, (size_t)H * sizeof(int), s));
    dim3 g(grid_for((long long)nchunks * 64, 256)), b(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_finalize_cached<double>, g, b, 0, s, (double*)filtered, rms, surveyMask, psMask, edgeCheck, snOut, maskOut, H,
                           W, apodPix, mulSurvey, chunkRange, nchunks, zeroRows);
    else
        hipLaunchKernelGGL(k_finalize_cached<float>, g, b, 0, s, (float*)filtered, rms, surveyMask, psMask, edgeCheck, snOut, maskOut, H, W,
                           apodPix, mulSurvey, chunkRange, nchunks, zeroRows);
    NF_CHECK_LAUNCH("k_finalize_cached");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// zero-restore after the pixel-window deconvolution (nemo/filters.py:99-101): out = (ref == 0) ? 0 : in
template <typename VT>
__global__ void k_restore_zeros(VT* __restrict__ data, const VT* __restrict__ ref, long long n) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x)
        if (ref[p] == (VT)0) data[p] = (VT)0;
}

int restore_zeros(int vtype, void* data, const void* ref, long long n, hipStream_t s) {
    dim3 g(grid_for(n, 256)), b(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_restore_zeros<double>, g, b, 0, s, (double*)data, (const double*)ref, n);
    else
        hipLaunchKernelGGL(k_restore_zeros<float>, g, b, 0, s, (float*)data, (const float*)ref, n);
    NF_CHECK_LAUNCH("k_restore_zeros");
    return NF_OK;
}

}  // namespace nf

// Batched 2-D FFTs for nemo tile shapes, fp32 and fp64, hand-written for gfx950.
//
// Replaces enmap.fft / enmap.ifft (pixell) at nemo/filters.py:525,548,559,615,646,662,679,847 and the
// transforms inside enmap.apply_window (nemo/filters.py:100,644).
//
// Structure: a 2-D transform is two "line" passes (rows, then columns, or the reverse for the inverse).
// One workgroup stages a bundle of lines in LDS (rows: whole contiguous rows; columns: a tile of
// adjacent columns so that every global access is a run of contiguous bytes), runs an in-place
// mixed-radix Cooley-Tukey on them (radices 2,3,4,5,7,8,11,13; digit-reversed placement is folded into
// the global->LDS load, so no second LDS buffer is needed), and writes them back.  Lengths with a
// larger prime factor (1031 and 1032 = 8*3*43 are real nemo tile dimensions) go through Bluestein's
// chirp-z algorithm, still inside LDS: DIF forward (natural -> digit-reversed), pointwise multiply by the
// pre-permuted chirp spectrum, DIT back (digit-reversed -> natural).
//
// The inverse direction uses conj(FFT(conj(x))), applied at load/store, so only forward twiddles exist.
#include "nf_common.h"
#include "nf_fft.h"

// The library is built with -ffp-contract=off (the stencil / painter kernels reproduce numpy roundings bit for bit);
// the butterflies gain accuracy and speed from fused multiply-adds, so contraction is switched back on here.
#pragma clang fp contract(fast)

#include <cmath>
#include <cstring>
#include <vector>

// ------------------------------------------------------------------------------------------------------
// device engine
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int digit_reverse(int i, const nf::LinePlan<T>& p) {
    int pos = 0, rem = i, blk = p.m;
    for (int j = p.nfac - 1; j >= 0; --j) {
        int r = p.fac[j];
        blk /= r;
        pos += (rem % r) * blk;
        rem /= r;
    }
    return pos;
}

template <typename T, int R>
struct Dft;

template <typename T>
struct Dft<T, 2> {
    static __device__ __forceinline__ void run(cpx<T>* a, const cpx<T>*) {
        cpx<T> t = a[1];
        a[1] = csub(a[0], t);
        a[0] = cadd(a[0], t);
    }
};
template <typename T>
struct Dft<T, 4> {
    static __device__ __forceinline__ void run(cpx<T>* a, const cpx<T>*) {
        cpx<T> t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
        cpx<T> t2 = cadd(a[1], a[3]), t3 = cmulnegi(csub(a[1], a[3]));
        a[0] = cadd(t0, t2);
        a[1] = cadd(t1, t3);
        a[2] = csub(t0, t2);
        a[3] = csub(t1, t3);
    }
};
template <typename T>
struct Dft<T, 8> {
    static __device__ __forceinline__ void run(cpx<T>* a, const cpx<T>*) {
        cpx<T> e[4] = {a[0], a[2], a[4], a[6]};
        cpx<T> o[4] = {a[1], a[3], a[5], a[7]};
        Dft<T, 4>::run(e, nullptr);
        Dft<T, 4>::run(o, nullptr);
        const T h = (T)0.70710678118654752440084436210485;
        cpx<T> w1o, w3o;
        w1o.x = (o[1].x + o[1].y) * h;  // o1 * (1 - i)/sqrt2
        w1o.y = (o[1].y - o[1].x) * h;
        w3o.x = (o[3].y - o[3].x) * h;  // o3 * (-1 - i)/sqrt2
        w3o.y = -(o[3].x + o[3].y) * h;
        cpx<T> w2o = cmulnegi(o[2]);
        a[0] = cadd(e[0], o[0]);
        a[4] = csub(e[0], o[0]);
        a[1] = cadd(e[1], w1o);
        a[5] = csub(e[1], w1o);
        a[2] = cadd(e[2], w2o);
        a[6] = csub(e[2], w2o);
        a[3] = cadd(e[3], w3o);
        a[7] = csub(e[3], w3o);
    }
};
template <typename T>
struct Dft<T, 3> {
    static __device__ __forceinline__ void run(cpx<T>* a, const cpx<T>*) {
        const T s = (T)0.86602540378443864676372317075294;
        cpx<T> t = cadd(a[1], a[2]);
        cpx<T> d = csub(a[1], a[2]);
        cpx<T> m1;
        m1.x = a[0].x - (T)0.5 * t.x;
        m1.y = a[0].y - (T)0.5 * t.y;
        cpx<T> m2;  // -i * s * d
        m2.x = s * d.y;
        m2.y = -s * d.x;
        a[0] = cadd(a[0], t);
        a[1] = cadd(m1, m2);
        a[2] = csub(m1, m2);
    }
};
// generic odd radix using the R-th roots wR[j] = exp(-2 pi i j / R)
template <typename T, int R>
struct Dft {
    static __device__ __forceinline__ void run(cpx<T>* a, const cpx<T>* wR) {
        constexpr int Hh = (R - 1) / 2;
        cpx<T> s[Hh + 1], d[Hh + 1];
#pragma unroll
        for (int j = 1; j <= Hh; ++j) {
            s[j] = cadd(a[j], a[R - j]);
            d[j] = csub(a[j], a[R - j]);
        }
        cpx<T> a0 = a[0];
        cpx<T> sum = a0;
#pragma unroll
        for (int j = 1; j <= Hh; ++j) sum = cadd(sum, s[j]);
        a[0] = sum;
#pragma unroll
        for (int k = 1; k <= Hh; ++k) {
            T re = a0.x, im = a0.y, re2 = 0, im2 = 0;
#pragma unroll
            for (int j = 1; j <= Hh; ++j) {
                cpx<T> w = wR[(j * k) % R];
                T c = w.x, sn = -w.y;
                re += s[j].x * c;
                im += s[j].y * c;
                re2 += d[j].y * sn;
                im2 += d[j].x * sn;
            }
            a[k].x = re + re2;
            a[k].y = im - im2;
            a[R - k].x = re - re2;
            a[R - k].y = im + im2;
        }
    }
};

// One in-place stage over all lines of the bundle.  DIT: twiddle inputs then butterfly; DIF: butterfly then twiddle.
template <typename T, int R, bool DIT>
__device__ __forceinline__ void fft_stage(cpx<T>* buf, const nf::LinePlan<T>& p, int L, int nl, int lpw, bool lineFast) {
    const int m = p.m;
    const int nb = m / R;
    const int Lp = L * R;
    const int twstep = m / Lp;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.tw[(long long)j * (m / R)];
    }
    const int total = nb * nl;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        int b, c;
        if (lineFast) {
            c = idx % nl;
            b = idx / nl;
        } else {
            b = idx % nb;
            c = idx / nb;
        }
        const int q = b % L, g = b / L;
        const int base = g * Lp + q;
        cpx<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            int e = base + k * L;
            a[k] = buf[lineFast ? e * lpw + c : c * m + e];
        }
        if (DIT && L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], p.tw[q * k * twstep]);
        }
        Dft<T, R>::run(a, wR);
        if (!DIT && L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) a[k] = cmul(a[k], p.tw[q * k * twstep]);
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            int e = base + k * L;
            buf[lineFast ? e * lpw + c : c * m + e] = a[k];
        }
    }
    __syncthreads();
}

template <typename T, bool DIT>
__device__ __forceinline__ void fft_stage_dispatch(cpx<T>* buf, const nf::LinePlan<T>& p, int r, int L, int nl, int lpw,
                                                   bool lineFast) {
    switch (r) {
        case 2: fft_stage<T, 2, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 3: fft_stage<T, 3, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 4: fft_stage<T, 4, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 5: fft_stage<T, 5, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 7: fft_stage<T, 7, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 8: fft_stage<T, 8, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 11: fft_stage<T, 11, DIT>(buf, p, L, nl, lpw, lineFast); break;
        case 13: fft_stage<T, 13, DIT>(buf, p, L, nl, lpw, lineFast); break;
        default: break;
    }
}

// digit-reversed input -> natural output
template <typename T>
__device__ void fft_dit(cpx<T>* buf, const nf::LinePlan<T>& p, int nl, int lpw, bool lineFast) {
    int L = 1;
    for (int j = 0; j < p.nfac; ++j) {
        fft_stage_dispatch<T, true>(buf, p, p.fac[j], L, nl, lpw, lineFast);
        L *= p.fac[j];
    }
}
// natural input -> digit-reversed output
template <typename T>
__device__ void fft_dif(cpx<T>* buf, const nf::LinePlan<T>& p, int nl, int lpw, bool lineFast) {
    int L = p.m;
    for (int j = p.nfac - 1; j >= 0; --j) {
        L /= p.fac[j];
        fft_stage_dispatch<T, false>(buf, p, p.fac[j], L, nl, lpw, lineFast);
    }
}

template <typename T>
__device__ __forceinline__ cpx<T> load_elem(const nf::LineArgs<T>& a, long long off, int line, int i) {
    cpx<T> v;
    if (a.inType == 0) {
        v = ((const cpx<T>*)a.in)[off];
    } else {
        double x = (a.inType == 1) ? ((const double*)a.in)[off] : (double)((const float*)a.in)[off];
        if (a.apodLine != nullptr) {
            // enmap.apod: axis -2 profile first, then axis -1, each multiply rounded to the map dtype
            double fl = a.apodLine[line], fe = a.apodElem[i];
            if (a.lineIsRow == 0) {
                double t = fl;
                fl = fe;
                fe = t;
            }
            if (a.inType == 2) {
                x = (double)(float)(x * fl);
                x = (double)(float)(x * fe);
            } else {
                x = x * fl;
                x = x * fe;
            }
        }
        v.x = (T)x;
        v.y = (T)0;
    }
    if (a.conjIn) v.y = -v.y;
    return v;
}

template <typename T, bool LINEFAST>
__global__ void k_fft_lines(nf::LineArgs<T> a) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<T>* buf = (cpx<T>*)nf_smem;
    const nf::LinePlan<T>& p = a.plan;
    const int n = p.n, m = p.m, lpw = a.linesPerWG;
    constexpr bool lineFast = LINEFAST;
    const int line0 = blockIdx.x * lpw;
    const int nl = min(lpw, a.nLines - line0);
    const long long inBase = (long long)blockIdx.y * a.inBatchStride;
    const long long outBase = (long long)blockIdx.y * a.outBatchStride;

    // ---- load (digit-reversed placement unless Bluestein) ----
    const int totalIn = (p.bluestein ? m : n) * nl;
    for (int idx = threadIdx.x; idx < totalIn; idx += blockDim.x) {
        int i, c;
        if (lineFast) {
            c = idx % nl;
            i = idx / nl;
        } else {
            i = idx % (p.bluestein ? m : n);
            c = idx / (p.bluestein ? m : n);
        }
        cpx<T> v;
        if (i < n) {
            long long off = inBase + (long long)(line0 + c) * a.inLineStride + (long long)i * a.inElemStride;
            v = load_elem<T>(a, off, line0 + c, i);
            if (p.bluestein) v = cmul(v, p.chirp[i]);
        } else {
            v.x = 0;
            v.y = 0;
        }
        int pos = p.bluestein ? i : digit_reverse<T>(i, p);
        buf[lineFast ? pos * lpw + c : c * m + pos] = v;
    }
    __syncthreads();

    if (!p.bluestein) {
        fft_dit<T>(buf, p, nl, lpw, lineFast);
    } else {
        fft_dif<T>(buf, p, nl, lpw, lineFast);
        const int tot = m * nl;
        for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
            int i, c;
            if (lineFast) {
                c = idx % nl;
                i = idx / nl;
            } else {
                i = idx % m;
                c = idx / m;
            }
            int at = lineFast ? i * lpw + c : c * m + i;
            buf[at] = cconj(cmul(buf[at], p.chirpFp[i]));
        }
        __syncthreads();
        fft_dit<T>(buf, p, nl, lpw, lineFast);
    }

    // ---- store ----
    const int totalOut = n * nl;
    for (int idx = threadIdx.x; idx < totalOut; idx += blockDim.x) {
        int i, c;
        if (lineFast) {
            c = idx % nl;
            i = idx / nl;
        } else {
            i = idx % n;
            c = idx / n;
        }
        cpx<T> v = buf[lineFast ? i * lpw + c : c * m + i];
        if (p.bluestein) v = cmul(cconj(v), p.chirp[i]);
        if (a.conjOut) v.y = -v.y;
        if (a.scaleDiv) {
            v.x = (T)(v.x / (T)a.scale);
            v.y = (T)(v.y / (T)a.scale);
        }
        long long off = outBase + (long long)(line0 + c) * a.outLineStride + (long long)i * a.outElemStride;
        if (a.outType == 0)
            ((cpx<T>*)a.out)[off] = v;
        else if (a.outType == 1)
            ((double*)a.out)[off] = (double)v.x;
        else
            ((float*)a.out)[off] = (float)v.x;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side: plans
// ------------------------------------------------------------------------------------------------------
namespace nf {

static bool factorize(int n, int* fac, int* nfac) {
    static const int radices[] = {8, 4, 2, 3, 5, 7, 11, 13};
    int k = 0;
    // keep radix-8/4 for the power-of-two part, but put small strides first so that early stages (small L)
    // are the cheap ones
    for (int r : radices) {
        while (n % r == 0) {
            if (k >= NF_MAXFAC) return false;
            fac[k++] = r;
            n /= r;
        }
    }
    *nfac = k;
    return n == 1;
}

static int next_smooth(int n) {
    for (int c = n;; ++c) {
        int t = c;
        for (int r : {2, 3, 5})
            while (t % r == 0) t /= r;
        if (t == 1) return c;
    }
}

template <typename T>
static int upload(const std::vector<cpx<T>>& v, const cpx<T>** dst) {
    void* d = nullptr;
    NF_HIP(hipMalloc(&d, v.size() * sizeof(cpx<T>)));
    NF_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(cpx<T>), hipMemcpyHostToDevice));
    *dst = (const cpx<T>*)d;
    return NF_OK;
}

static int host_digit_reverse(int i, int m, const int* fac, int nfac) {
    int pos = 0, rem = i, blk = m;
    for (int j = nfac - 1; j >= 0; --j) {
        int r = fac[j];
        blk /= r;
        pos += (rem % r) * blk;
        rem /= r;
    }
    return pos;
}

// plain O(m log m) host FFT (long double) used only to build the Bluestein chirp spectrum
static void host_fft(std::vector<long double>& re, std::vector<long double>& im) {
    const int m = (int)re.size();
    // naive mixed approach: recursive on smallest prime factor
    if (m == 1) return;
    int r = 2;
    while (m % r != 0) ++r;
    const int sub = m / r;
    std::vector<std::vector<long double>> sre(r, std::vector<long double>(sub)), sim(r, std::vector<long double>(sub));
    for (int i = 0; i < m; ++i) {
        sre[i % r][i / r] = re[i];
        sim[i % r][i / r] = im[i];
    }
    for (int k = 0; k < r; ++k) host_fft(sre[k], sim[k]);
    const long double pi2 = 6.283185307179586476925286766559L;
    for (int f = 0; f < m; ++f) {
        long double ar = 0, ai = 0;
        for (int k = 0; k < r; ++k) {
            long long e = ((long long)f * k) % m;
            long double ang = -pi2 * (long double)e / (long double)m;
            long double c = cosl(ang), s = sinl(ang);
            long double xr = sre[k][f % sub], xi = sim[k][f % sub];
            ar += xr * c - xi * s;
            ai += xr * s + xi * c;
        }
        re[f] = ar;
        im[f] = ai;
    }
}

template <typename T>
int make_line_plan(int n, LinePlan<T>* p) {
    memset(p, 0, sizeof(*p));
    p->n = n;
    const long double pi = 3.141592653589793238462643383279502884L;
    if (factorize(n, p->fac, &p->nfac)) {
        p->m = n;
        p->bluestein = 0;
    } else {
        p->bluestein = 1;
        p->m = next_smooth(2 * n - 1);
        if (!factorize(p->m, p->fac, &p->nfac)) {
            nf_set_error("internal: smooth length failed to factorize");
            return NF_ERR_SIZE;
        }
    }
    const int m = p->m;
    std::vector<cpx<T>> tw(m);
    for (int k = 0; k < m; ++k) {
        long double ang = -2.0L * pi * (long double)k / (long double)m;
        tw[k].x = (T)cosl(ang);
        tw[k].y = (T)sinl(ang);
    }
    int rc = upload<T>(tw, &p->tw);
    if (rc) return rc;
    if (p->bluestein) {
        std::vector<cpx<T>> chirp(n);
        std::vector<long double> bre(m, 0.0L), bim(m, 0.0L);
        for (int k = 0; k < n; ++k) {
            long long k2 = ((long long)k * k) % (2LL * n);
            long double ang = pi * (long double)k2 / (long double)n;
            chirp[k].x = (T)cosl(ang);
            chirp[k].y = (T)(-sinl(ang));
            bre[k] = cosl(ang);
            bim[k] = sinl(ang);
            if (k > 0) {
                bre[m - k] = cosl(ang);
                bim[m - k] = sinl(ang);
            }
        }
        host_fft(bre, bim);
        std::vector<cpx<T>> cf(m);
        for (int f = 0; f < m; ++f) {
            int pos = host_digit_reverse(f, m, p->fac, p->nfac);
            cf[pos].x = (T)(bre[f] / (long double)m);
            cf[pos].y = (T)(bim[f] / (long double)m);
        }
        rc = upload<T>(chirp, &p->chirp);
        if (rc) return rc;
        rc = upload<T>(cf, &p->chirpFp);
        if (rc) return rc;
    }
    return NF_OK;
}

template <typename T>
void free_line_plan(LinePlan<T>* p) {
    if (p->tw) (void)hipFree((void*)p->tw);
    if (p->chirp) (void)hipFree((void*)p->chirp);
    if (p->chirpFp) (void)hipFree((void*)p->chirpFp);
    memset(p, 0, sizeof(*p));
}

static const int kMaxLds = 160 * 1024;

template <typename T>
int launch_lines(const LineArgs<T>& a, int batch, int threads, hipStream_t stream) {
    size_t lds = (size_t)a.plan.m * a.linesPerWG * sizeof(cpx<T>);
    if (lds > (size_t)kMaxLds) {
        nf_set_error("FFT line of this length does not fit in LDS");
        return NF_ERR_SIZE;
    }
    static bool attrSet[2] = {false, false};
    int which = sizeof(T) == 8 ? 1 : 0;
    if (!attrSet[which]) {
        NF_HIP(hipFuncSetAttribute((const void*)k_fft_lines<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        NF_HIP(hipFuncSetAttribute((const void*)k_fft_lines<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
        attrSet[which] = true;
    }
    const double npixTotal = (double)a.nLines * a.plan.n * batch;
    const double inB = a.inType == 0 ? sizeof(cpx<T>) : (a.inType == 1 ? 8.0 : 4.0);
    const double outB = a.outType == 0 ? sizeof(cpx<T>) : (a.outType == 1 ? 8.0 : 4.0);
    const int tag = sizeof(T) == 8 ? (a.lineFast ? PT_FFT_COLS_F64 : PT_FFT_ROWS_F64) : (a.lineFast ? PT_FFT_COLS_F32 : PT_FFT_ROWS_F32);
    ProfScope ps(tag, stream, npixTotal * (inB + outB));
    dim3 grid(nf_div_up(a.nLines, a.linesPerWG), batch);
    if (a.lineFast)
        hipLaunchKernelGGL((k_fft_lines<T, true>), grid, dim3(threads), lds, stream, a);
    else
        hipLaunchKernelGGL((k_fft_lines<T, false>), grid, dim3(threads), lds, stream, a);
    NF_CHECK_LAUNCH("k_fft_lines");
    return NF_OK;
}

template <typename T>
int Plan2D<T>::init(int H_, int W_) {
    H = H_;
    W = W_;
    int rc = make_line_plan<T>(W, &row);
    if (rc) return rc;
    rc = make_line_plan<T>(H, &col);
    if (rc) return rc;
    const size_t es = sizeof(cpx<T>);
    // rows: bundle enough rows for ~32 KB of LDS, at least one
    size_t rowBytes = (size_t)row.m * es;
    if (rowBytes > (size_t)kMaxLds) {
        nf_set_error("row length too large for the single-pass LDS FFT");
        return NF_ERR_SIZE;
    }
    rowsPerWG = (int)std::max<size_t>(1, std::min<size_t>(8, (32 * 1024) / rowBytes));
    // columns: as many adjacent columns as fit in ~64 KB (8 * 16 B = 128 B contiguous when possible)
    size_t colBytes = (size_t)col.m * es;
    if (colBytes > (size_t)kMaxLds) {
        nf_set_error("column length too large for the single-pass LDS FFT");
        return NF_ERR_SIZE;
    }
    size_t want = 128 / es;  // columns for a 128-byte run
    size_t fit = (size_t)kMaxLds / colBytes;
    size_t target = std::max<size_t>(1, (64 * 1024) / colBytes);
    colsPerWG = (int)std::max<size_t>(1, std::min<size_t>(fit, std::max<size_t>(std::min<size_t>(want, fit), std::min<size_t>(target, 16))));
    threadsRow = (row.m * rowsPerWG >= 2048) ? 256 : 128;
    if ((size_t)row.m * rowsPerWG * es > 48 * 1024) threadsRow = 512;
    threadsCol = (col.m * colsPerWG >= 2048) ? 256 : 128;
    if ((size_t)col.m * colsPerWG * es > 48 * 1024) threadsCol = 512;
    return NF_OK;
}

template <typename T>
void Plan2D<T>::destroy() {
    free_line_plan<T>(&row);
    free_line_plan<T>(&col);
}

// forward: in (real f32/f64 or complex<T>) -> out complex<T> [batch][H][W]; optional apodisation of real input;
// result divided by `scale` if scaleDiv (enmap.fft: sqrt(H*W)).
template <typename T>
int Plan2D<T>::forward(const void* in, int inType, void* out, int batch, const double* apodY, const double* apodX,
                       int scaleDiv, double scale, hipStream_t stream) {
    LineArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.plan = row;
    a.in = in;
    a.out = out;
    a.inType = inType;
    a.outType = 0;
    a.inElemStride = 1;
    a.inLineStride = W;
    a.inBatchStride = (long long)H * W;
    a.outElemStride = 1;
    a.outLineStride = W;
    a.outBatchStride = (long long)H * W;
    a.nLines = H;
    a.linesPerWG = rowsPerWG;
    a.lineFast = 0;
    a.lineIsRow = 1;
    a.apodLine = apodY;
    a.apodElem = apodX;
    int rc = launch_lines<T>(a, batch, threadsRow, stream);
    if (rc) return rc;
    LineArgs<T> c;
    memset(&c, 0, sizeof(c));
    c.plan = col;
    c.in = out;
    c.out = out;
    c.inType = 0;
    c.outType = 0;
    c.inElemStride = W;
    c.inLineStride = 1;
    c.inBatchStride = (long long)H * W;
    c.outElemStride = W;
    c.outLineStride = 1;
    c.outBatchStride = (long long)H * W;
    c.nLines = W;
    c.linesPerWG = colsPerWG;
    c.lineFast = 1;
    c.scaleDiv = scaleDiv;
    c.scale = scale;
    return launch_lines<T>(c, batch, threadsCol, stream);
}

// inverse (unnormalised backward sum, then optional division by `scale`): complex<T> in -> work (complex, same
// size, may alias `in` if the caller allows clobbering) -> out (complex<T>, or real part as f64 / f32).
template <typename T>
int Plan2D<T>::inverse(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                       hipStream_t stream) {
    LineArgs<T> c;
    memset(&c, 0, sizeof(c));
    c.plan = col;
    c.in = in;
    c.out = work;
    c.inType = 0;
    c.outType = 0;
    c.inElemStride = W;
    c.inLineStride = 1;
    c.inBatchStride = (long long)H * W;
    c.outElemStride = W;
    c.outLineStride = 1;
    c.outBatchStride = (long long)H * W;
    c.nLines = W;
    c.linesPerWG = colsPerWG;
    c.lineFast = 1;
    c.conjIn = 1;  // conj(FFT(conj x)): the conj between the two passes cancels, so only ends matter
    int rc = launch_lines<T>(c, batch, threadsCol, stream);
    if (rc) return rc;
    LineArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.plan = row;
    a.in = work;
    a.out = out;
    a.inType = 0;
    a.outType = outType;
    a.inElemStride = 1;
    a.inLineStride = W;
    a.inBatchStride = (long long)H * W;
    a.outElemStride = 1;
    a.outLineStride = W;
    a.outBatchStride = (long long)H * W;
    a.nLines = H;
    a.linesPerWG = rowsPerWG;
    a.lineFast = 0;
    a.lineIsRow = 1;
    a.conjOut = 1;
    a.scaleDiv = scaleDiv;
    a.scale = scale;
    return launch_lines<T>(a, batch, threadsRow, stream);
}

template struct Plan2D<float>;
template struct Plan2D<double>;

}  // namespace nf

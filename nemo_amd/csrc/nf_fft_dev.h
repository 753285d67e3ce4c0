// Device-side pieces shared by the line-transform kernels (nf_fft.hip, nf_linefilt.hip): the small DFTs, the LDS swizzle
// and the per-thread geometry record.  Included after the translation unit has chosen its fp-contract mode.
#pragma once
#include "nf_common.h"

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

__device__ __forceinline__ int swz(int e) {
    return e ^ (int)(__builtin_amdgcn_ubfe((unsigned)e, 3u, 5u) ^ __builtin_amdgcn_ubfe((unsigned)e, 9u, 5u));
}

// what a thread needs to know about where its lines live
struct Geo {
    int nl;               // valid lines of this workgroup
    int tb, TB;           // butterfly lane and stride
    int tl, TL;           // rows kernel: line lane and stride;  columns kernel: tl = column within the tile, TL = tile width
    int line0;            // first global line of the workgroup
    const void* inP;      // a.in / a.out advanced to this workgroup's plane (offsets inside a plane fit 32 bits)
    void* outP;
};

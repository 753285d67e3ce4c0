// Internal C++ interface of the FFT module (see nf_fft.hip).
#pragma once
#include "nf_common.h"

#define NF_MAXFAC 24

namespace nf {

template <typename T>
struct LinePlan {
    int n;          // logical transform length
    int m;          // working length in LDS (== n, or the padded Bluestein length)
    int bluestein;  // 1 if n has a prime factor > 13
    int nfac;
    int fac[NF_MAXFAC];     // radices of m, applied in this order by the DIT pass
    const cpx<T>* tw;       // [m]  exp(-2 pi i k / m)
    const cpx<T>* chirp;    // [n]  exp(-i pi k^2 / n)                        (Bluestein only)
    const cpx<T>* chirpFp;  // [m]  FFT_m(conj chirp, wrapped) / m, digit-reversed positions (Bluestein only)
};

template <typename T>
struct LineArgs {
    LinePlan<T> plan;
    const void* in;
    void* out;
    int inType, outType;  // 0: complex<T>; 1: real f64; 2: real f32 (output = real part)
    long long inElemStride, inLineStride, inBatchStride;
    long long outElemStride, outLineStride, outBatchStride;
    int nLines;
    int linesPerWG;
    int lineFast;   // LDS layout: 1 = [element][line] (column tiles), 0 = [line][element] (rows)
    int lineIsRow;  // which apod table applies to the line index (1: apodLine is the axis -2 profile)
    int conjIn, conjOut;
    int scaleDiv;
    double scale;
    const double* apodLine;  // optional per-line apodisation factors (length nLines)
    const double* apodElem;  // optional per-element apodisation factors (length n)
};

template <typename T>
struct Plan2D {
    int H, W;
    LinePlan<T> row, col;
    int rowsPerWG, colsPerWG, threadsRow, threadsCol;
    int init(int H, int W);
    void destroy();
    int forward(const void* in, int inType, void* out, int batch, const double* apodY, const double* apodX,
                int scaleDiv, double scale, hipStream_t stream);
    int inverse(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                hipStream_t stream);
};

}  // namespace nf

// Internal C++ interface of the FFT module (see nf_fft.hip).
#pragma once
#include "nf_common.h"
#include <vector>

#define NF_MAXFAC 24

namespace nf {

// One butterfly stage of a line transform: radix R on elements base + k*L (k < R), base = (b / L) * L * R + b % L.
struct StageInfo {
    int R;
    int L;           // product of the radices of the earlier (DIT order) stages
    unsigned magic;  // ceil(2^32 / L): b / L == __umulhi(b, magic) for every b, L < 2^15   (unused when L == 1)
    int twOff;       // stage twiddles in LinePlan::stw: [(k-1) * L + q] = exp(-2 pi i q k / (L R)), k = 1..R-1, q < L
    int rootOff;     // R-th roots of unity in LinePlan::stw (odd radices > 3 only)
};

template <typename T>
struct LinePlan {
    int n;          // logical transform length
    int m;          // working length in LDS (== n, or the padded Bluestein length)
    int mPad;       // m rounded up to a multiple of 32 (the LDS swizzle permutes within 32-element blocks)
    int bluestein;  // 1 if n has a prime factor > 13
    int nst;
    StageInfo st[NF_MAXFAC];    // DIT order: st[0].L == 1
    const cpx<T>* stw;          // stage twiddles and small-root tables
    const unsigned short* rev;  // [m] digit-reversed position of natural index i
    const unsigned short* src;  // [m] natural index held at digit-reversed position e (inverse of rev)
    const cpx<T>* chirp;        // [n]  exp(-i pi k^2 / n)                        (Bluestein only)
    const cpx<T>* chirpFp;      // [m]  FFT_m(conj chirp, wrapped) / m, digit-reversed positions (Bluestein only)
};

#define NF_PLANE_SCALES 16      // room of the per-plane factor table of a transform (a power of two)

template <typename T>
struct LineArgs {
    LinePlan<T> plan;
    const void* in;
    void* out;
    int inType, outType;  // 0: complex<T>; 1: real f64; 2: real f32 (output = real part)
    long long inElemStride, inLineStride, inBatchStride;
    long long outElemStride, outLineStride, outBatchStride;
    int nLines;
    int linesPerWG;  // rows kernel: lines staged per workgroup; columns kernel: adjacent columns per tile (power of two)
    int tb;          // rows kernel: threads along the butterfly index (the rest of the workgroup strides over lines)
    int nTiles;      // columns kernel: number of column tiles
    double conjInSign, conjOutSign;  // -1 conjugates the complex input / output (inverse = conj FFT conj), else +1
    double scale;    // multiplier applied at the store (1 / sqrt(H*W) for unitary transforms, else 1)
    // pointwise work fused into the columns kernel's load (inverse transforms): see fused_ops() in nf_fft.hip
    int fused;               // any of the following present
    const float* mulF32;     // float32 filter, indexed like the input plane
    const double* mulSym;    // halfIn: the symmetrised filter 0.5 (filt[k] + filt[-k]) in float64, half-stored like the input (instead of mulF32)
    long long mulBatchStride;  // filter plane offset per batch entry (elements; 0 = the same filter for every batch entry)
    int nSum;                // planes summed as sum_i in_i * mul_i (1 = none)
    long long sumStride;     // plane pitch of in / mul for the sum (elements)
    int storeLo, storeHi;    // columns kernel: only the elements (rows) storeLo <= e < storeHi are stored (storeHi == 0: all)
    int lineBase;            // rows kernel: first line of this launch (lines lineBase .. nLines - 1)
    int rowSwap64;           // rows kernel, pairMode 2: row y of the input plane lies at row ((y & 63) << 6) | (y >> 6) (H = 4096;
                             //   what the two-pass column transform k_cols2p_* leaves)
    int colPair;             // columns kernel, float32: strides, types and table alignment allow two adjacent columns per thread
    int usePlaneScale;       // halfIn: plane pl of the sum (nSum > 1) / batch entry b (nSum <= 1) is also multiplied by planeScale[pl / b]
    double planeScale[NF_PLANE_SCALES];
    const double* winLine;   // pixel-window table along the line index (columns kernel: wx[col]) and
    const double* winElem;   //   along the element index (wy[row]); both NULL = no window
    // real-data transforms (half the work of a complex transform; see "Real input / real output" in nf_fft.hip)
    int pairMode;            // rows kernel: 0 = one line per row; 1 = forward, a line is the complex pair (row 2p) + i (row 2p+1)
                             //   of REAL input rows, unpacked to two half-rows; 2 = inverse, two Hermitian half-rows are packed
                             //   into one complex line whose real / imaginary parts are the two REAL output rows
    int mirror;              // columns kernel, forward: column c of the half plane is also stored, conjugated and reversed, as
                             //   column W - c (the other half of the spectrum of a real map)
    int herm;                // columns kernel, inverse: load 0.5 * (G[y][c] + conj G[-y][-c]) (real part of the output only)
    int halfIn;              //   ... of a spectrum of a real map that is stored for c <= W/2 only (the mirror mode is the
                             //   conjugate of the stored one; only its filter value and window factor are fetched)
    int realH, realW;        // shape of the real map (H rows, W columns)
    int packNyq;             // columns kernel: lane 0 of tile 0 carries columns 0 and W/2 as one complex column (nLines = W/2)
    // rows kernel, pairMode 2 (real output): optional epilogues on the value being stored
    const double* epiScale;          // device scalar: out = (float)value * scale (* epiMask), formed in the output type
    long long epiScaleStride;        //   batch entry b takes epiScale[b * epiScaleStride]
    const unsigned char* epiMask;    //   mask multiplied in after the scale
    const void* epiZeroRef;          // map of the output type: where it is zero the output is zero
    const int* epiZeroRows;          // line filter, rows: per map row, 0 = that row of epiZeroRef holds no zero (it is not read)
    const double* apodLine;  // optional per-line apodisation factors (length nLines); rows kernel, real input only
    const double* apodElem;  // optional per-element apodisation factors (length n)
};

// nf_linefilt.hip: the rows line filter with fused first / middle / last stages (plain radices up to 8, >= 2 stages)
template <typename T>
int launch_line_filter_rows_fused(const LineArgs<T>& a, int batch, int threads, size_t lds, hipStream_t stream);
template <typename T>
int launch_line_filter_cols_fused(const LineArgs<T>& a, int batch, int threads, size_t lds, hipStream_t stream);

// The template path of one matched-filter scale with ONE frequency map as a single column pass (k_template_cols): what the
// column kernel needs beside the two transforms' own arguments
#define NF_TF_MAXE 16          // elements of a tile a thread carries through the pointwise step (n * tile width / threads)
#define NF_TF_MAXBATCH 8
struct TemplateFuse {
    const void* cov;           // [H][W] smoothed noise power of the one map (all scales of the batch)
    double w0;                 // its frequency weight
    float* filt[NF_TF_MAXBATCH];   // per batch entry: the float32 filter plane [H][W], or NULL (not wanted)
    double* sym;               // [batch][H][W] symmetrised filter, half-stored (written)
};

template <typename T>
struct Plan2D {
    int H, W;
    LinePlan<T> row, col;
    int rowsPerWG, colsPerWG, threadsRow, tbRow, threadsCol;
    int useReal;    // real-data fast paths enabled (NF_FFT_REAL=0 switches them off)
    const cpx<T>* tw4096 = nullptr;   // H == 4096: exp(-2 pi i k / 4096), the twiddles of the two-pass column transform (else NULL)
    int packNyq;    // columns 0 and W/2 of half spectra share one complex column (NF_FFT_PACK_NYQ=0 switches it off)
    std::vector<unsigned short> revRow, revCol;   // host copies of row.rev / col.rev (digit-reversed position of index k)
    bool packOk() const;
    int axis_filter(int axis, const void* in, int inType, void* out, int outType, int batch, const double* wperm,
                    const void* zeroRef, hipStream_t stream, const int* zeroRows = nullptr);
    int init(int H, int W);
    void destroy();
    int forward(const void* in, int inType, void* out, int batch, const double* apodY, const double* apodX,
                int scaleDiv, double scale, hipStream_t stream, int half = 0, int rowsOnly = 0);
    // can template_filter run for this plan (float32 plain-radix columns, a thread's share of a tile within NF_TF_MAXE)?
    bool template_filter_ok(int batch) const;
    // rowsOut: what forward(..., half = 1, rowsOnly = 1) left in the spectrum buffer.  Column transform, filter build
    // (k_build_filter_sym, N = 1), symmetrised filter x planeScale x window, inverse column transform into `work`, then the inverse
    // row pass into `out` - the values of forward() + build_filter_sym() + inverse_half() bit for bit; the spectrum itself is
    // never stored.
    int template_filter(const void* rowsOut, int batch, const TemplateFuse& tf, int covType, const double* planeScale, const double* winY,
                        const double* winX, void* work, void* out, int outType, int rowLo, int rowHi, int scaleDiv, double scale,
                        hipStream_t stream);
    int inverse(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                hipStream_t stream, const float* filt = nullptr, int nSum = 1, long long filtBatchStride = 0,
                const double* winY = nullptr, const double* winX = nullptr);
    // inverse of half-stored spectra (real output): filtSym = symmetrised float64 filter planes, half-stored (or NULL)
    int inverse_half(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                     hipStream_t stream, const double* filtSym, int nSum, long long filtBatchStride, const double* winY,
                     const double* winX, const double* epiScale = nullptr, const unsigned char* epiMask = nullptr,
                     const void* epiZeroRef = nullptr, const double* planeScale = nullptr, int rowLo = 0, int rowHi = 0,
                     long long inBatchStride = -1, long long epiScaleStride = 0);
};

}  // namespace nf

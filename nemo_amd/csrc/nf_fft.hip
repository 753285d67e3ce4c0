// Batched 2-D FFTs for nemo tile shapes, fp32 and fp64, hand-written for gfx950.
//
// Replaces enmap.fft / enmap.ifft (pixell) at nemo/filters.py:525,548,559,615,646,662,679,847 and the
// transforms inside enmap.apply_window (nemo/filters.py:100,644).
//
// Structure: a 2-D transform is two "line" passes (rows, then columns, or the reverse for the inverse), each
// reading and writing the plane once.  A workgroup keeps a bundle of lines in LDS and runs an in-place
// mixed-radix decimation-in-time FFT on it (radices 2,3,4,5,7,8,11,13):
//   * rows kernel: whole contiguous rows; the digit-reversed placement is folded into the global -> LDS copy
//     (table lookup), the last stage stores straight from registers to global memory (natural order,
//     consecutive lanes -> consecutive elements);
//   * columns kernel: a tile of adjacent columns, so that every global access is a run of contiguous bytes;
//     the FIRST stage loads its operands straight from global memory (row src[e] for digit-reversed position
//     e - for a column tile any row order is equally coalesced) and the LAST stage stores straight to global
//     memory, so a transform of S stages crosses LDS only S-1 times; tiles are dealt to the XCDs so that
//     neighbouring tiles (which share 128-byte lines when the tile is narrow) run on the same XCD / L2.
// Stage twiddles come from per-stage tables laid out [k][q] (consecutive lanes read consecutive entries and a
// thread reuses them for every line it owns); b / L and b % L use a multiply-high by a precomputed reciprocal;
// LDS element e lives at e ^ (((e >> 3) ^ (e >> 9)) & 31), which keeps the stride-1, stride-8 and stride-64/512
// access patterns of the radix-8 stages free of bank conflicts without padding.
// Lengths with a prime factor > 13 (1031 and 1032 = 8*3*43 are real nemo tile dimensions) go through
// Bluestein's chirp-z algorithm inside LDS: DIF forward (natural -> digit-reversed), pointwise multiply by the
// pre-permuted chirp spectrum, DIT back (digit-reversed -> natural).
//
// The inverse direction uses conj(FFT(conj(x))), applied at load/store, so only forward twiddles exist.
#include "nf_common.h"
#include "nf_fft.h"

// The library is built with -ffp-contract=off (the stencil / painter kernels reproduce numpy roundings bit for bit);
// the butterflies gain accuracy and speed from fused multiply-adds, so contraction is switched back on here.
#pragma clang fp contract(fast)

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NF_FFT_MAXTHREADS 1024
#define NF_STAGE_IN 8

// ------------------------------------------------------------------------------------------------------
// device engine
// ------------------------------------------------------------------------------------------------------
#include "nf_fft_dev.h"

// IT: 0 complex<T>, 1 real float64, 2 real float32, 3 / 4 the same two with apodisation (rows kernel only: `fl` is the
// axis -2 profile value of the row, apodElem the axis -1 profile).  OT: 0 complex<T>, 1 real float64, 2 real float32.
// Compile-time so that the per-element code has no branches.
template <typename T, int IT>
__device__ __forceinline__ cpx<T> load_elem(const nf::LineArgs<T>& a, const void* inP, unsigned off, double fl, int i) {
    cpx<T> v;
    if (IT == 0) {
        v = ((const cpx<T>*)inP)[off];
        v.y *= (T)a.conjInSign;
    } else {
        constexpr bool F32 = (IT == 2 || IT == 4);
        double x = F32 ? (double)((const float*)inP)[off] : ((const double*)inP)[off];
        if (IT >= 3) {
            // enmap.apod: axis -2 profile first, then axis -1, each multiply rounded to the map dtype
            const double fe = a.apodElem[i];
            if (F32) {
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
    return v;
}

template <typename T, int OT, bool BLUE>
__device__ __forceinline__ void store_elem(const nf::LineArgs<T>& a, void* outP, unsigned off, int i, cpx<T> v) {
    const nf::LinePlan<T>& p = a.plan;
    if (BLUE) v = cmul(cconj(v), p.chirp[i]);
    // multiplication by the rounded reciprocal of sqrt(H*W): within 1 ulp of the reference's division, exact when
    // H*W is a power of four; 1 when the transform is not scaled
    const T r = (T)a.scale;
    if (OT == 0 || OT == 3) {
        v.x *= r;
        v.y *= r * (T)a.conjOutSign;
        ((cpx<T>*)outP)[off] = v;
    } else if (OT == 1) {
        ((double*)outP)[off] = (double)(v.x * r);
    } else {
        ((float*)outP)[off] = (float)(v.x * r);
    }
}

// Pointwise work fused into the column kernel's global load (the pass that would otherwise read and write the whole
// plane once more): pixel-window factor (enmap.apply_window, nemo/filters.py:100,644: product of the two 1-D tables
// formed first, applied in float64, rounded to T) and / or the float32 filter with the frequency sum
// (nemo/filters.py:847, sum moved in front of the inverse transform).  `line` = column, `row` = element index of the
// mode the factors belong to; `off` = where the further planes of the sum are read (imaginary parts times imSign),
// `offMul` = where the filter is read (differs from `off` only for the mirror mode of a half-stored spectrum).
template <typename T>
__device__ __forceinline__ cpx<T> fused_ops(const nf::LineArgs<T>& a, const Geo& G, cpx<T> v, unsigned off, T imSign,
                                             unsigned offMul, int line, int row) {
    double win = 1.0;
    if (a.winElem != nullptr) {
        win = a.winElem[row] * a.winLine[line];
        v.x = (T)((double)v.x * win);
        v.y = (T)((double)v.y * win);
    }
    if (a.mulF32 != nullptr) {
        const float* mul = a.mulF32 + (size_t)blockIdx.y * (size_t)a.mulBatchStride;
        T h = (T)mul[offMul];
        T re = v.x * h, im = v.y * h;
        for (int pl = 1; pl < a.nSum; ++pl) {
            cpx<T> f = ((const cpx<T>*)G.inP)[(size_t)pl * (size_t)a.sumStride + off];
            f.y *= (T)a.conjInSign * imSign;
            if (a.winElem != nullptr) {
                f.x = (T)((double)f.x * win);
                f.y = (T)((double)f.y * win);
            }
            h = (T)mul[(size_t)pl * (size_t)a.sumStride + offMul];
            re += f.x * h;
            im += f.y * h;
        }
        v.x = re;
        v.y = im;
    }
    return v;
}

// Inverse transform of which only the real part is wanted: real(ifft2(G)) = ifft2(0.5 * (G[k] + conj G[-k])), and the
// symmetrised spectrum needs only its columns c <= W/2 transformed (the rows pass rebuilds the rest by conjugation).
// `v` is the (conjugated-for-the-inverse) element at (row, col) after load_elem; the mirror element goes through the same
// load and the same pointwise factors, then u = 0.5 * (u1 + conj u2).
template <typename T>
__device__ __forceinline__ cpx<T> fused_ops_herm(const nf::LineArgs<T>& a, const Geo& G, cpx<T> v, unsigned off, int col, int row) {
    if (a.herm && a.halfIn) {
        // Half-stored spectra of real maps: the mirror mode is the conjugate of the stored one, the window tables are even,
        // so 0.5 (G[k] + conj G[-k]) = win[k] * sum_i M_i[k] * filtSym_i[k] with the symmetrised filter - one real factor
        // per plane, formed in float64.
        double re = 0.0, im = 0.0;
        const double* __restrict__ ms = a.mulSym ? a.mulSym + (size_t)blockIdx.y * (size_t)a.mulBatchStride : nullptr;
        const int np_ = a.nSum > 1 ? a.nSum : 1;
        for (int pl = 0; pl < np_; ++pl) {
            cpx<T> f = v;
            if (pl > 0) {
                f = ((const cpx<T>*)G.inP)[(size_t)pl * (size_t)a.sumStride + off];
                f.y *= (T)a.conjInSign;
            }
            double g = ms ? ms[(size_t)pl * (size_t)a.sumStride + off] : 1.0;
            if (a.usePlaneScale) g *= a.planeScale[a.nSum > 1 ? pl : (int)(blockIdx.y & (NF_PLANE_SCALES - 1))];
            re += (double)f.x * g;
            im += (double)f.y * g;
        }
        if (a.winElem != nullptr) {
            const double win = a.winElem[row] * a.winLine[col];
            re *= win;
            im *= win;
        }
        cpx<T> h;
        h.x = (T)re;
        h.y = (T)im;
        return h;
    }
    cpx<T> u1 = fused_ops<T>(a, G, v, off, (T)1, off, col, row);
    if (!a.herm) return u1;
    const int rowM = row == 0 ? 0 : a.realH - row;
    const int colM = col == 0 ? 0 : a.realW - col;
    const unsigned offM = (unsigned)colM * (unsigned)a.inLineStride + (unsigned)rowM * (unsigned)a.inElemStride;
    const cpx<T> u2 = fused_ops<T>(a, G, load_elem<T, 0>(a, G.inP, offM, 1.0, rowM), offM, (T)1, offM, colM, rowM);
    cpx<T> h;
    h.x = (T)0.5 * (u1.x + u2.x);
    h.y = (T)0.5 * (u1.y - u2.y);
    return h;
}

// One in-place stage.  DIT (dif == false): twiddle the inputs, then the butterfly; DIF: butterfly, then twiddle the
// outputs.  srcG: operands come from global memory (columns kernel, first DIT stage: row src[e]); dstG: results go
// to global memory (last DIT stage: natural index e).
// SPEC (compile time, so that the common kernels do not carry the registers of the special paths): columns kernel
// 1 = mirrored second store (forward transform of a real map), 2 = pointwise / symmetrised load (fused inverse),
// 3 = one half-stored plane times a real factor (batched form of 2); the rows kernel always passes 0.
template <typename T, int R, bool COLS, bool BLUE, int SPEC>
__device__ __forceinline__ void fft_stage(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st,
                                          const bool dif, const bool srcG, const bool dstG, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R;
    const unsigned nb = (unsigned)p.m / R;
    const unsigned lineMul = COLS ? 1u : (unsigned)p.mPad;   // LDS index = line * lineMul + swz(e) * elemMul
    const unsigned elemMul = COLS ? (unsigned)G.TL : 1u;
    const cpx<T>* __restrict__ twp = p.stw + st.twOff;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    for (unsigned b = (unsigned)G.tb; b < nb; b += (unsigned)G.TB) {
        unsigned g = b, q = 0;
        if (L > 1) {
            g = __umulhi(b, st.magic);
            q = b - g * L;
        }
        const unsigned base = g * LR + q;
        cpx<T> w[R - 1];
        if (L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) w[k - 1] = twp[(unsigned)(k - 1) * L + q];
        }
        unsigned idx[R];   // LDS element index, or (srcG) the global row
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned e = base + (unsigned)k * L;
            idx[k] = srcG ? (unsigned)p.src[e] : (unsigned)swz((int)e) * elemMul;
        }
        for (int line = G.tl; line < G.nl; line += (COLS ? G.nl : G.TL)) {
            cpx<T>* __restrict__ lb = buf + (unsigned)line * lineMul;
            cpx<T> x[R];
            if (srcG) {
                const unsigned lo = (unsigned)(G.line0 + line) * (unsigned)a.inLineStride;
#pragma unroll
                for (int k = 0; k < R; ++k)
                    x[k] = load_elem<T, 0>(a, G.inP, lo + idx[k] * (unsigned)a.inElemStride, 1.0, (int)idx[k]);
                if (COLS && SPEC == 2) {      // after the plain loads, so that those stay batched
#pragma unroll
                    for (int k = 0; k < R; ++k)
                        x[k] = fused_ops_herm<T>(a, G, x[k], lo + idx[k] * (unsigned)a.inElemStride, G.line0 + line, (int)idx[k]);
                }
                if (COLS && a.packNyq && G.line0 + line == 0) {
                    // lane 0 of tile 0 carries columns 0 and W/2 as ONE complex column z0 + i zN (see k_fft_cols): fetch the
                    // Nyquist column's operands the general way and fold them in.  Forward: both columns are real.
                    const int colN = a.realW / 2;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const unsigned offN = (unsigned)colN * (unsigned)a.inLineStride + idx[k] * (unsigned)a.inElemStride;
                        cpx<T> zN = load_elem<T, 0>(a, G.inP, offN, 1.0, (int)idx[k]);
                        if (SPEC >= 2) {
                            zN = fused_ops_herm<T>(a, G, zN, offN, colN, (int)idx[k]);
                            if (SPEC == 3) x[k] = fused_ops_herm<T>(a, G, x[k], lo + idx[k] * (unsigned)a.inElemStride, 0, (int)idx[k]);
                        } else {
                            x[k].y = (T)0;
                            zN.y = (T)0;
                        }
                        x[k].x -= zN.y;
                        x[k].y += zN.x;
                    }
                } else if (COLS && SPEC == 3) {
                    // fused_ops_herm for ONE half-stored plane (the matched filter's own case): the real factor
                    // filtSym[k] * win[k] per element, its operands fetched in one batch like the data
                    const int col = G.line0 + line;
                    double f[R];
                    if (a.mulSym != nullptr) {
                        const double* __restrict__ ms = a.mulSym + (size_t)blockIdx.y * (size_t)a.mulBatchStride;
#pragma unroll
                        for (int k = 0; k < R; ++k) f[k] = ms[lo + idx[k] * (unsigned)a.inElemStride];
                    } else {
#pragma unroll
                        for (int k = 0; k < R; ++k) f[k] = 1.0;
                    }
                    if (a.usePlaneScale) {
                        const double ps = a.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)];
#pragma unroll
                        for (int k = 0; k < R; ++k) f[k] *= ps;
                    }
                    if (a.winElem != nullptr) {
                        const double wl = a.winLine[col];
#pragma unroll
                        for (int k = 0; k < R; ++k) f[k] *= a.winElem[idx[k]] * wl;
                    }
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        x[k].x = (T)((double)x[k].x * f[k]);
                        x[k].y = (T)((double)x[k].y * f[k]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) x[k] = lb[idx[k]];
            }
            if (L > 1 && !dif) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
            Dft<T, R>::run(x, wR);
            if (L > 1 && dif) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
            if (dstG) {
                const unsigned lo = (unsigned)(G.line0 + line) * (unsigned)a.outLineStride;
#define NF_STORE_ALL(OT)                                                                                          \
    _Pragma("unroll") for (int k = 0; k < R; ++k) {                                                               \
        const unsigned e = base + (unsigned)k * L;                                                                \
        if ((!BLUE || e < (unsigned)p.n) && (!COLS || a.storeHi == 0 || (e >= (unsigned)a.storeLo && e < (unsigned)a.storeHi))) \
            store_elem<T, OT, BLUE>(a, G.outP, lo + e * (unsigned)a.outElemStride, (int)e, x[k]); \
    }
                if (COLS && SPEC >= 2 && a.packNyq && G.line0 + line == 0) {
                    // inverse of the packed column: the two real columns come back as real and imaginary part
                    const T r = (T)a.scale, ri = r * (T)a.conjOutSign;
                    const unsigned colN = (unsigned)(a.realW / 2);
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const unsigned e = base + (unsigned)k * L;
                        cpx<T> v0, vN;
                        v0.x = x[k].x * r;
                        v0.y = (T)0;
                        vN.x = x[k].y * ri;
                        vN.y = (T)0;
                        if (a.storeHi == 0 || (e >= (unsigned)a.storeLo && e < (unsigned)a.storeHi)) {
                            ((cpx<T>*)G.outP)[e * (unsigned)a.outElemStride] = v0;
                            ((cpx<T>*)G.outP)[e * (unsigned)a.outElemStride + colN] = vN;
                        }
                    }
                } else if (COLS && SPEC == 1) {
                    // spectrum of a real map: F[-y][-c] = conj F[y][c]; this tile holds columns c <= W/2 and writes both
                    NF_STORE_ALL(0)
                    const unsigned col = (unsigned)(G.line0 + line);
                    if (col > 0u && 2u * col < (unsigned)a.realW) {
                        const T r = (T)a.scale;
#pragma unroll
                        for (int k = 0; k < R; ++k) {
                            const unsigned e = base + (unsigned)k * L;
                            if (!BLUE || e < (unsigned)p.n) {
                                cpx<T> v = x[k];
                                if (BLUE) v = cmul(cconj(v), p.chirp[e]);
                                v.x *= r;
                                v.y *= -(r * (T)a.conjOutSign);
                                const unsigned rowM = e == 0u ? 0u : (unsigned)a.realH - e;
                                ((cpx<T>*)G.outP)[rowM * (unsigned)a.outElemStride + ((unsigned)a.realW - col)] = v;
                            }
                        }
                    }
                } else if (a.outType == 0) {
                    NF_STORE_ALL(0)
                } else if (a.outType == 1) {
                    NF_STORE_ALL(1)
                } else {
                    NF_STORE_ALL(2)
                }
#undef NF_STORE_ALL
            } else if (srcG) {
#pragma unroll
                for (int k = 0; k < R; ++k) lb[(unsigned)swz((int)(base + (unsigned)k * L)) * elemMul] = x[k];
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) lb[idx[k]] = x[k];
            }
        }
    }
    if (!dstG) __syncthreads();
}

// BIG == false: instantiates only the radices up to 8 (fewer registers: every nemo tile length without a factor 11 or
// 13 runs this variant)
template <typename T, bool COLS, int MODE, int SPEC>
__device__ __forceinline__ void fft_stage_dispatch(cpx<T>* buf, const nf::LineArgs<T>& a, int s, bool dif, bool srcG,
                                                   bool dstG, const Geo& G) {
    const nf::StageInfo st = a.plan.st[s];
    constexpr bool BIG = MODE == 1;
    if (!BIG && st.R > 8) return;
    switch (st.R) {
        case 2: fft_stage<T, 2, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 3: fft_stage<T, 3, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 4: fft_stage<T, 4, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 5: fft_stage<T, 5, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 7: fft_stage<T, 7, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 8: fft_stage<T, 8, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 11: if (BIG) fft_stage<T, 11, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        case 13: if (BIG) fft_stage<T, 13, COLS, MODE == 2, SPEC>(buf, a, st, dif, srcG, dstG, G); break;
        default: break;
    }
}

// Columns kernel, float32, first (srcG) or last (!srcG) stage with TWO adjacent columns per thread: the pair travels as one
// 16-byte access to global memory and to LDS (one complex64 per lane makes 8-byte accesses and 32-byte runs per tile row), the
// twiddles are fetched once for both.  Same arithmetic per element as fft_stage (DIT, plain radices, complex in and out, SPEC
// 0 or 3); the caller takes this path only for whole tiles of even width away from the packed column 0 / W/2, with even
// pitches and 16-byte aligned planes.
template <int R, int SPEC>
__device__ __forceinline__ void fft_stage_colpair(cpx<float>* __restrict__ buf, const nf::LineArgs<float>& a, const nf::StageInfo st,
                                                  const bool srcG, const Geo& G) {
    const nf::LinePlan<float>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R;
    const unsigned nb = (unsigned)p.m / R;
    const unsigned TL = (unsigned)G.TL, P = TL >> 1;
    const unsigned pl = threadIdx.x % P, tb2 = threadIdx.x / P, TB2 = blockDim.x / P;
    const unsigned line = 2u * pl, col = (unsigned)G.line0 + line;
    const cpx<float>* __restrict__ twp = p.stw + st.twOff;
    cpx<float> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    for (unsigned b = tb2; b < nb; b += TB2) {
        unsigned g = b, q = 0;
        if (L > 1) {
            g = __umulhi(b, st.magic);
            q = b - g * L;
        }
        const unsigned base = g * LR + q;
        cpx<float> w[R - 1];
        if (L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) w[k - 1] = twp[(unsigned)(k - 1) * L + q];
        }
        cpx<float> x0[R], x1[R];
        if (srcG) {
            unsigned row[R];
#pragma unroll
            for (int k = 0; k < R; ++k) row[k] = (unsigned)p.src[base + (unsigned)k * L];
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float4 v = *(const float4*)((const cpx<float>*)G.inP + (col + row[k] * (unsigned)a.inElemStride));
                x0[k].x = v.x;
                x0[k].y = v.y * (float)a.conjInSign;
                x1[k].x = v.z;
                x1[k].y = v.w * (float)a.conjInSign;
            }
            if (SPEC == 3) {
                // the real factor filtSym * planeScale * (wy * wx) per element, formed in fft_stage's order
                double2 m2[R];
                if (a.mulSym != nullptr) {
                    const double* __restrict__ ms = a.mulSym + (size_t)blockIdx.y * (size_t)a.mulBatchStride;
#pragma unroll
                    for (int k = 0; k < R; ++k) m2[k] = *(const double2*)(ms + (col + row[k] * (unsigned)a.inElemStride));
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        m2[k].x = 1.0;
                        m2[k].y = 1.0;
                    }
                }
                const double ps = a.usePlaneScale ? a.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)] : 1.0;
                const bool win = a.winElem != nullptr;
                const double wl0 = win ? a.winLine[col] : 1.0, wl1 = win ? a.winLine[col + 1] : 1.0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    double f0 = m2[k].x, f1 = m2[k].y;
                    if (a.usePlaneScale) {
                        f0 *= ps;
                        f1 *= ps;
                    }
                    if (win) {
                        const double we = a.winElem[row[k]];
                        f0 *= we * wl0;
                        f1 *= we * wl1;
                    }
                    x0[k].x = (float)((double)x0[k].x * f0);
                    x0[k].y = (float)((double)x0[k].y * f0);
                    x1[k].x = (float)((double)x1[k].x * f1);
                    x1[k].y = (float)((double)x1[k].y * f1);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const float4 v = *(const float4*)(buf + ((unsigned)swz((int)(base + (unsigned)k * L)) * TL + line));
                x0[k].x = v.x;
                x0[k].y = v.y;
                x1[k].x = v.z;
                x1[k].y = v.w;
            }
        }
        if (L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) {
                x0[k] = cmul(x0[k], w[k - 1]);
                x1[k] = cmul(x1[k], w[k - 1]);
            }
        }
        Dft<float, R>::run(x0, wR);
        Dft<float, R>::run(x1, wR);
        if (srcG) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                float4 v;
                v.x = x0[k].x;
                v.y = x0[k].y;
                v.z = x1[k].x;
                v.w = x1[k].y;
                *(float4*)(buf + ((unsigned)swz((int)(base + (unsigned)k * L)) * TL + line)) = v;
            }
        } else {
            const float r = (float)a.scale, ri = r * (float)a.conjOutSign;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const unsigned e = base + (unsigned)k * L;
                if (a.storeHi == 0 || (e >= (unsigned)a.storeLo && e < (unsigned)a.storeHi)) {
                    float4 v;
                    v.x = x0[k].x * r;
                    v.y = x0[k].y * ri;
                    v.z = x1[k].x * r;
                    v.w = x1[k].y * ri;
                    *(float4*)((cpx<float>*)G.outP + (col + e * (unsigned)a.outElemStride)) = v;
                }
            }
        }
    }
    if (srcG) __syncthreads();
}

template <int SPEC>
__device__ __forceinline__ bool fft_stage_colpair_dispatch(cpx<float>* buf, const nf::LineArgs<float>& a, int s, bool srcG, const Geo& G) {
    const nf::StageInfo st = a.plan.st[s];
    switch (st.R) {
        case 2: fft_stage_colpair<2, SPEC>(buf, a, st, srcG, G); return true;
        case 3: fft_stage_colpair<3, SPEC>(buf, a, st, srcG, G); return true;
        case 4: fft_stage_colpair<4, SPEC>(buf, a, st, srcG, G); return true;
        case 5: fft_stage_colpair<5, SPEC>(buf, a, st, srcG, G); return true;
        case 7: fft_stage_colpair<7, SPEC>(buf, a, st, srcG, G); return true;
        case 8: fft_stage_colpair<8, SPEC>(buf, a, st, srcG, G); return true;
        default: return false;
    }
}

// Bluestein middle: natural order data in LDS -> DIF -> multiply by the chirp spectrum (conjugated: the way back is a
// forward transform of the conjugate) -> ready for the DIT stages.
template <typename T, bool COLS, int MODE>
__device__ __forceinline__ void bluestein_middle(cpx<T>* buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    for (int s = p.nst - 1; s >= 0; --s) fft_stage_dispatch<T, COLS, MODE, 0>(buf, a, s, true, false, false, G);
    for (int e = G.tb; e < p.m; e += G.TB) {
        const cpx<T> cf = p.chirpFp[e];
        for (int line = G.tl; line < G.nl; line += (COLS ? G.nl : G.TL)) {
            const int at = COLS ? swz(e) * G.TL + line : line * p.mPad + swz(e);
            buf[at] = cconj(cmul(buf[at], cf));
        }
    }
    __syncthreads();
}

template <typename T, int IT, bool BLUE>
__device__ __forceinline__ void stage_in_rows(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const int nIn = BLUE ? p.m : p.n;
    for (int line = G.tl; line < G.nl; line += G.TL) {
        const unsigned lineOff = (unsigned)(G.line0 + line) * (unsigned)a.inLineStride;
        cpx<T>* __restrict__ lb = buf + line * p.mPad;
        const double fl = IT >= 3 ? a.apodLine[G.line0 + line] : 1.0;
        for (int i0 = G.tb; i0 < nIn; i0 += NF_STAGE_IN * G.TB) {
            // branch-free load phase (indices clamped to the line) so that all loads are in flight together
            cpx<T> v[NF_STAGE_IN], ch[BLUE ? NF_STAGE_IN : 1];
            int pos[NF_STAGE_IN];
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = min(i0 + j * G.TB, p.n - 1);
                v[j] = load_elem<T, IT>(a, G.inP, lineOff + (unsigned)i * (unsigned)a.inElemStride, fl, i);
                if (BLUE)
                    ch[j] = p.chirp[i];
                else
                    pos[j] = p.rev[i];
            }
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = i0 + j * G.TB;
                if (i < nIn) {
                    cpx<T> t = v[j];
                    if (BLUE) {
                        t = cmul(t, ch[j]);
                        if (i >= p.n) {
                            t.x = 0;
                            t.y = 0;
                        }
                        lb[swz(i)] = t;
                    } else {
                        lb[swz(pos[j])] = t;
                    }
                }
            }
        }
    }
}

// Real input / real output (rows kernel, pairMode).
//  forward (pairMode 1): line p = (row 2p) + i (row 2p+1) of the REAL map; after the transform Z,
//      F_a[k] = 0.5 (Z[k] + conj Z[W-k]),  F_b[k] = -0.5 i (Z[k] - conj Z[W-k]),   k = 0 .. W/2,
//      are the half-rows of the two spectra (the columns pass completes the plane by conjugate symmetry).
//  inverse (pairMode 2): the two Hermitian half-rows A, B (k <= W/2; the rest is conj of the mirror; the imaginary parts
//      of k = 0 and k = W/2 are rounding noise and are dropped) are packed as Z = A + i B; the inverse transform of Z
//      has the two real output rows as its real and imaginary parts.
template <typename T, int IT, bool BLUE>
__device__ __forceinline__ void stage_in_pairs_fwd(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const int nIn = BLUE ? p.m : p.n;
    for (int line = G.tl; line < G.nl; line += G.TL) {
        const int rowA = 2 * (G.line0 + line);
        const bool haveB = rowA + 1 < a.realH;
        const int rowB = haveB ? rowA + 1 : rowA;
        const unsigned offA = (unsigned)rowA * (unsigned)a.realW, offB = (unsigned)rowB * (unsigned)a.realW;
        cpx<T>* __restrict__ lb = buf + line * p.mPad;
        const double flA = IT >= 3 ? a.apodLine[rowA] : 1.0, flB = IT >= 3 ? a.apodLine[rowB] : 1.0;
        for (int i0 = G.tb; i0 < nIn; i0 += NF_STAGE_IN * G.TB) {
            cpx<T> v[NF_STAGE_IN], ch[BLUE ? NF_STAGE_IN : 1];
            int pos[NF_STAGE_IN];
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = min(i0 + j * G.TB, p.n - 1);
                v[j].x = load_elem<T, IT>(a, G.inP, offA + (unsigned)i, flA, i).x;
                v[j].y = load_elem<T, IT>(a, G.inP, offB + (unsigned)i, flB, i).x;
                if (BLUE)
                    ch[j] = p.chirp[i];
                else
                    pos[j] = p.rev[i];
            }
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = i0 + j * G.TB;
                if (i < nIn) {
                    cpx<T> t = v[j];
                    if (!haveB) t.y = 0;
                    if (BLUE) {
                        t = cmul(t, ch[j]);
                        if (i >= p.n) {
                            t.x = 0;
                            t.y = 0;
                        }
                        lb[swz(i)] = t;
                    } else {
                        lb[swz(pos[j])] = t;
                    }
                }
            }
        }
    }
}

template <typename T, bool BLUE>
__device__ __forceinline__ void unpack_pairs_fwd(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const int n = p.n, nh = n / 2 + 1;
    const T r = (T)a.scale * (T)0.5;
    for (int line = G.tl; line < G.nl; line += G.TL) {
        const int rowA = 2 * (G.line0 + line);
        const bool haveB = rowA + 1 < a.realH;
        const cpx<T>* __restrict__ lb = buf + line * p.mPad;
        cpx<T>* outA = (cpx<T>*)G.outP + (unsigned)rowA * (unsigned)a.realW;
        for (int k = G.tb; k < nh; k += G.TB) {
            const int km = k == 0 ? 0 : n - k;
            cpx<T> zk = lb[swz(k)], zm = lb[swz(km)];
            if (BLUE) {
                zk = cmul(cconj(zk), p.chirp[k]);
                zm = cmul(cconj(zm), p.chirp[km]);
            }
            cpx<T> fa, fb;
            fa.x = r * (zk.x + zm.x);
            fa.y = r * (zk.y - zm.y);
            fb.x = r * (zk.y + zm.y);
            fb.y = -r * (zk.x - zm.x);
            outA[k] = fa;
            if (haveB) outA[(unsigned)a.realW + k] = fb;
        }
    }
}

template <typename T, bool BLUE>
__device__ __forceinline__ void stage_in_pairs_inv(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const int n = p.n, nIn = BLUE ? p.m : p.n;
    const T sgn = (T)a.conjInSign;
    for (int line = G.tl; line < G.nl; line += G.TL) {
        const int rowA = 2 * (G.line0 + line);
        const bool haveB = rowA + 1 < a.realH;
        // (rowSwap64: the two-pass column transform left row y at row ((y & 63) << 6) | (y >> 6); rowA is even, so rowA + 1 lies 64 rows on)
        const unsigned physA = a.rowSwap64 ? (((unsigned)rowA & 63u) << 6) | ((unsigned)rowA >> 6) : (unsigned)rowA;
        const cpx<T>* inA = (const cpx<T>*)G.inP + physA * (unsigned)a.realW;
        const cpx<T>* inB = inA + (haveB ? (a.rowSwap64 ? 64u : 1u) * (unsigned)a.realW : 0u);
        cpx<T>* __restrict__ lb = buf + line * p.mPad;
        if (!BLUE) {
            // each stored mode k <= n/2 is loaded once and feeds both line elements k and n - k (the conjugate)
            const int nh = n >> 1;
            for (int k0 = G.tb; k0 <= nh; k0 += NF_STAGE_IN * G.TB) {
                cpx<T> va[NF_STAGE_IN], vb[NF_STAGE_IN];
                int p1[NF_STAGE_IN], p2[NF_STAGE_IN];
#pragma unroll
                for (int j = 0; j < NF_STAGE_IN; ++j) {
                    const int k = min(k0 + j * G.TB, nh);
                    va[j] = inA[k];
                    vb[j] = inB[k];
                    p1[j] = p.rev[k];
                    p2[j] = p.rev[k == 0 ? 0 : n - k];
                }
#pragma unroll
                for (int j = 0; j < NF_STAGE_IN; ++j) {
                    const int k = k0 + j * G.TB;
                    if (k <= nh) {
                        cpx<T> A = va[j], B = vb[j];
                        const bool self = k == 0 || 2 * k == n;      // self-conjugate modes are real
                        if (self) {
                            A.y = 0;
                            B.y = 0;
                        }
                        if (!haveB) {
                            B.x = 0;
                            B.y = 0;
                        }
                        cpx<T> t;
                        t.x = A.x - B.y;
                        t.y = (A.y + B.x) * sgn;
                        lb[swz(p1[j])] = t;
                        if (!self) {
                            t.x = A.x + B.y;
                            t.y = (B.x - A.y) * sgn;
                            lb[swz(p2[j])] = t;
                        }
                    }
                }
            }
            continue;
        }
        for (int i0 = G.tb; i0 < nIn; i0 += NF_STAGE_IN * G.TB) {
            cpx<T> va[NF_STAGE_IN], vb[NF_STAGE_IN], ch[BLUE ? NF_STAGE_IN : 1];
            int pos[NF_STAGE_IN];
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = min(i0 + j * G.TB, n - 1);
                const int k = 2 * i <= n ? i : n - i;
                va[j] = inA[k];
                vb[j] = inB[k];
                if (BLUE)
                    ch[j] = p.chirp[i];
                else
                    pos[j] = p.rev[i];
            }
#pragma unroll
            for (int j = 0; j < NF_STAGE_IN; ++j) {
                const int i = i0 + j * G.TB;
                if (i < nIn) {
                    const int ic = min(i, n - 1);
                    cpx<T> A = va[j], B = vb[j];
                    if (2 * ic > n) {              // mirrored half: conjugate
                        A.y = -A.y;
                        B.y = -B.y;
                    }
                    if (ic == 0 || 2 * ic == n) {  // self-conjugate modes are real
                        A.y = 0;
                        B.y = 0;
                    }
                    if (!haveB) {
                        B.x = 0;
                        B.y = 0;
                    }
                    cpx<T> t;
                    t.x = A.x - B.y;
                    t.y = (A.y + B.x) * sgn;
                    if (BLUE) {
                        t = cmul(t, ch[j]);
                        if (i >= n) {
                            t.x = 0;
                            t.y = 0;
                        }
                        lb[swz(i)] = t;
                    } else {
                        lb[swz(pos[j])] = t;
                    }
                }
            }
        }
    }
}

// Inverse of a real map's half-stored rows, last step: the complex line in LDS holds row 2p in its real and row 2p+1 in its
// imaginary part (of the conjugated transform).  Consecutive threads take consecutive pixels: coalesced stores, and the
// optional epilogues of the matched-filter path applied to the real value on its way out:
//   epiScale    float32 map * signalNorm * psMask (nemo/filters.py:847,853,714: the filtered map is cast to float32, then
//               multiplied by the scalar and the mask in the output type)
//   epiZeroRef  pixels that are zero in the reference map stay zero (nemo/filters.py:99-101)
template <typename T, bool BLUE>
__device__ __forceinline__ void store_pairs_inv(const cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const int n = p.n;
    const T r = (T)a.scale, rb = r * (T)a.conjOutSign;
    const bool doScale = a.epiScale != nullptr;
    const double es = doScale ? a.epiScale[(size_t)blockIdx.y * (size_t)a.epiScaleStride] : 1.0;
    const bool useMask = doScale && a.epiMask != nullptr, useRef = a.epiZeroRef != nullptr;
    constexpr int NB = 4;
    for (int line = G.tl; line < G.nl; line += G.TL) {
        const unsigned rowA = 2u * (unsigned)(G.line0 + line);
        const bool haveB = rowA + 1u < (unsigned)a.realH;
        const cpx<T>* __restrict__ lb = buf + line * p.mPad;
        const unsigned oa0 = rowA * (unsigned)a.realW;
        for (int e0 = G.tb; e0 < n; e0 += NB * G.TB) {
            // operands of NB pixels first (one batch of loads): bits 0-7 mask of row A, 8-15 mask of row B, 16 / 17
            // "reference is zero" for row A / B
            cpx<T> z[NB];
            unsigned ep[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int ec = min(e0 + j * G.TB, n - 1);
                z[j] = lb[swz(ec)];
                unsigned v = 0x0101u;
                if (useMask || useRef) {
                    const unsigned oa = oa0 + (unsigned)ec, ob = haveB ? oa + (unsigned)a.realW : oa;
                    if (useMask) v = (unsigned)a.epiMask[oa] | ((unsigned)a.epiMask[ob] << 8);
                    if (useRef) {
                        if (a.outType == 1) {
                            v |= ((const double*)a.epiZeroRef)[oa] == 0.0 ? 0x10000u : 0u;
                            v |= ((const double*)a.epiZeroRef)[ob] == 0.0 ? 0x20000u : 0u;
                        } else {
                            v |= ((const float*)a.epiZeroRef)[oa] == 0.0f ? 0x10000u : 0u;
                            v |= ((const float*)a.epiZeroRef)[ob] == 0.0f ? 0x20000u : 0u;
                        }
                    }
                }
                ep[j] = v;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int e = e0 + j * G.TB;
                if (e < n) {
                    cpx<T> v = z[j];
                    if (BLUE) v = cmul(cconj(v), p.chirp[e]);
                    const unsigned oa = oa0 + (unsigned)e, ob = oa + (unsigned)a.realW;
                    T va = v.x * r, vb = v.y * rb;
                    if (a.outType == 1) {
                        double wa = (double)va, wb = (double)vb;
                        if (doScale) {
                            wa = (double)(float)va * es;
                            wb = (double)(float)vb * es;
                            if (useMask) {
                                wa = wa * (double)(ep[j] & 0xffu);
                                wb = wb * (double)((ep[j] >> 8) & 0xffu);
                            }
                        }
                        if (ep[j] & 0x10000u) wa = 0.0;
                        if (ep[j] & 0x20000u) wb = 0.0;
                        ((double*)G.outP)[oa] = wa;
                        if (haveB) ((double*)G.outP)[ob] = wb;
                    } else {
                        float wa = (float)va, wb = (float)vb;
                        if (doScale) {
                            wa = wa * (float)es;
                            wb = wb * (float)es;
                            if (useMask) {
                                wa = wa * (float)(ep[j] & 0xffu);
                                wb = wb * (float)((ep[j] >> 8) & 0xffu);
                            }
                        }
                        if (ep[j] & 0x10000u) wa = 0.0f;
                        if (ep[j] & 0x20000u) wb = 0.0f;
                        ((float*)G.outP)[oa] = wa;
                        if (haveB) ((float*)G.outP)[ob] = wb;
                    }
                }
            }
        }
    }
}

// Rows: lines are contiguous in memory (elemStride 1).  LDS layout [line][swz(e)].
// (the plain-radix variants are held to 128 VGPRs - four wavefronts per SIMD - so that two 512-thread workgroups share a
// CU and one's butterflies overlap the other's HBM traffic; the big-radix and Bluestein variants keep their registers)
template <typename T, int MODE, int PAIR>
__global__ void __launch_bounds__(512, (MODE == 0 ? 4 : 2)) k_fft_rows(nf::LineArgs<T> a) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<T>* buf = (cpx<T>*)nf_smem;
    const nf::LinePlan<T>& p = a.plan;
    Geo G;
    G.line0 = a.lineBase + blockIdx.x * a.linesPerWG;
    G.nl = min(a.linesPerWG, a.nLines - G.line0);
    G.TB = a.tb;
    G.TL = blockDim.x / a.tb;
    G.tl = threadIdx.x / a.tb;
    G.tb = threadIdx.x - G.tl * a.tb;
    {
        const size_t inEs = a.inType == 0 ? sizeof(cpx<T>) : (a.inType == 1 ? 8 : 4);
        const size_t outEs = a.outType == 0 ? sizeof(cpx<T>) : (a.outType == 1 ? 8 : 4);
        G.inP = (const char*)a.in + (size_t)blockIdx.y * (size_t)a.inBatchStride * inEs;
        G.outP = (char*)a.out + (size_t)blockIdx.y * (size_t)a.outBatchStride * outEs;
    }

    // global -> LDS in batches of NF_STAGE_IN loads per thread, all issued before the first LDS store: the copy is
    // latency-bound otherwise (one HBM round trip per element and thread)
    const int it = a.inType == 0 ? 0 : a.inType + (a.apodLine != nullptr ? 2 : 0);
    if (PAIR == 1) {
        switch (it) {
            case 1: stage_in_pairs_fwd<T, 1, MODE == 2>(buf, a, G); break;
            case 2: stage_in_pairs_fwd<T, 2, MODE == 2>(buf, a, G); break;
            case 3: stage_in_pairs_fwd<T, 3, MODE == 2>(buf, a, G); break;
            default: stage_in_pairs_fwd<T, 4, MODE == 2>(buf, a, G); break;
        }
    } else if (PAIR == 2) {
        stage_in_pairs_inv<T, MODE == 2>(buf, a, G);
    } else {
        switch (it) {
            case 0: stage_in_rows<T, 0, MODE == 2>(buf, a, G); break;
            case 1: stage_in_rows<T, 1, MODE == 2>(buf, a, G); break;
            case 2: stage_in_rows<T, 2, MODE == 2>(buf, a, G); break;
            case 3: stage_in_rows<T, 3, MODE == 2>(buf, a, G); break;
            default: stage_in_rows<T, 4, MODE == 2>(buf, a, G); break;
        }
    }
    __syncthreads();
    if (MODE == 2) bluestein_middle<T, false, MODE>(buf, a, G);
    if (PAIR == 1) {
        // the unpacking needs Z[k] and Z[W-k] together: the last stage stays in LDS
        for (int s = 0; s < p.nst; ++s) fft_stage_dispatch<T, false, MODE, 0>(buf, a, s, false, false, false, G);
        unpack_pairs_fwd<T, MODE == 2>(buf, a, G);
        return;
    }
    if (PAIR == 2) {
        // real output rows: the last stage stays in LDS too, the rows leave through a coalesced store pass
        for (int s = 0; s < p.nst; ++s) fft_stage_dispatch<T, false, MODE, 0>(buf, a, s, false, false, false, G);
        store_pairs_inv<T, MODE == 2>(buf, a, G);
        return;
    }
    for (int s = 0; s < p.nst; ++s) fft_stage_dispatch<T, false, MODE, 0>(buf, a, s, false, false, s == p.nst - 1, G);
}

// Columns: a tile of TL = linesPerWG adjacent lines (elemStride = row pitch, lineStride 1).  LDS layout [swz(e)][line].
// packNyq (half-plane passes of real maps, W even): lane 0 of tile 0 carries columns 0 and W/2 as one complex column
// c0 + i cN - before the forward column pass both are real (row spectra at k = 0 and k = W/2), before the inverse one both
// are Hermitian along y, so their transforms separate again (forward: with the mirror element; inverse: real / imaginary
// part).  The pass then covers W/2 columns instead of W/2 + 1.
template <typename T, int MODE, int SPEC>
__device__ __forceinline__ void fft_cols_body(const nf::LineArgs<T>& a) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<T>* buf = (cpx<T>*)nf_smem;
    const nf::LinePlan<T>& p = a.plan;
    // blocks b, b + 8, b + 16 ... share an XCD (round-robin dispatch): give them neighbouring tiles
    const int per = (a.nTiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= a.nTiles) return;
    Geo G;
    G.TL = a.linesPerWG;
    G.line0 = tile * G.TL;
    const int nvalid = min(G.TL, a.nLines - G.line0);
    G.tl = threadIdx.x & (G.TL - 1);
    G.tb = threadIdx.x / G.TL;
    G.TB = blockDim.x / G.TL;
    // a lane whose column lies beyond the map takes no part (nl = 0 empties its line loops); barriers stay uniform
    G.nl = G.tl < nvalid ? G.tl + 1 : 0;
    {
        const size_t inEs = a.inType == 0 ? sizeof(cpx<T>) : (a.inType == 1 ? 8 : 4);
        const size_t outEs = a.outType == 0 ? sizeof(cpx<T>) : (a.outType == 1 ? 8 : 4);
        G.inP = (const char*)a.in + (size_t)blockIdx.y * (size_t)a.inBatchStride * inEs;
        G.outP = (char*)a.out + (size_t)blockIdx.y * (size_t)a.outBatchStride * outEs;
    }

    if (MODE == 2) {
        if (G.nl > 0) {
            for (int i = G.tb; i < p.m; i += G.TB) {
                cpx<T> v;
                if (i < p.n) {
                    const unsigned off = (unsigned)(G.line0 + G.tl) * (unsigned)a.inLineStride + (unsigned)i * (unsigned)a.inElemStride;
                    v = load_elem<T, 0>(a, G.inP, off, 1.0, i);
                    if (SPEC >= 2) v = fused_ops_herm<T>(a, G, v, off, G.line0 + G.tl, i);
                    v = cmul(v, p.chirp[i]);
                } else {
                    v.x = 0;
                    v.y = 0;
                }
                buf[swz(i) * G.TL + G.tl] = v;
            }
        }
        __syncthreads();
        bluestein_middle<T, true, MODE>(buf, a, G);
        for (int s = 0; s < p.nst; ++s) fft_stage_dispatch<T, true, MODE, SPEC>(buf, a, s, false, false, s == p.nst - 1, G);
    } else {
        // packNyq, forward: Z = FFT(c0 + i cN) is unpacked with its mirror element, G0[k] = (Z[k] + conj Z[-k]) / 2,
        // GN[k] = -i (Z[k] - conj Z[-k]) / 2 - tile 0 keeps its last stage in LDS and leaves through a store pass
        const bool unpack = SPEC <= 1 && a.packNyq && tile == 0;
        // float32, whole tiles away from the packed column: two adjacent columns per thread in the stages that touch global memory
        bool pairOK = false;
        if constexpr (sizeof(T) == 4 && MODE == 0 && (SPEC == 0 || SPEC == 3)) {
            pairOK = a.colPair && G.TL >= 2 && nvalid == G.TL && !(a.packNyq && tile == 0) && p.nst >= 2 && (blockDim.x % (G.TL >> 1)) == 0 &&
                     (((size_t)G.inP | (size_t)G.outP) & 15) == 0;
        }
        for (int s = 0; s < p.nst; ++s) {
            const bool srcG = s == 0, dstG = s == p.nst - 1 && !unpack;
            bool done = false;
            if constexpr (sizeof(T) == 4 && MODE == 0 && (SPEC == 0 || SPEC == 3)) {
                if (pairOK && (srcG || dstG)) done = fft_stage_colpair_dispatch<SPEC>((cpx<float>*)buf, a, s, srcG, G);
            }
            if (!done) fft_stage_dispatch<T, true, MODE, SPEC>(buf, a, s, false, srcG, dstG, G);
        }
        if (SPEC <= 1 && unpack) {
            const unsigned n = (unsigned)p.n, TL = (unsigned)G.TL, colN = (unsigned)(a.realW / 2);
            const T r = (T)a.scale, ri = r * (T)a.conjOutSign, h = (T)0.5;
            for (unsigned i = threadIdx.x; i < n * TL; i += blockDim.x) {
                const unsigned line = i & (TL - 1), e = i / TL;
                if ((int)line >= nvalid) continue;
                cpx<T> v = buf[(unsigned)swz((int)e) * TL + line];
                cpx<T>* orow = (cpx<T>*)G.outP + e * (unsigned)a.outElemStride;
                if (line == 0) {
                    const cpx<T> zm = buf[(unsigned)swz((int)(e == 0u ? 0u : n - e)) * TL];
                    cpx<T> g0, gN;
                    g0.x = h * (v.x + zm.x) * r;
                    g0.y = h * (v.y - zm.y) * ri;
                    gN.x = h * (v.y + zm.y) * r;
                    gN.y = -h * (v.x - zm.x) * ri;
                    orow[0] = g0;
                    orow[colN] = gN;
                } else {
                    v.x *= r;
                    v.y *= ri;
                    const unsigned col = (unsigned)G.line0 + line;
                    orow[col] = v;
                    if (SPEC == 1 && 2u * col < (unsigned)a.realW) {      // the other half of a real map's spectrum
                        v.y = -v.y;
                        ((cpx<T>*)G.outP)[(e == 0u ? 0u : (unsigned)a.realH - e) * (unsigned)a.outElemStride + ((unsigned)a.realW - col)] = v;
                    }
                }
            }
        }
    }
}


// Long columns run as one wide workgroup per CU (1024 threads: at most 128 registers per lane); short ones (the tiles of a
// survey, a few KB per column) as 256-thread workgroups, several per CU - compiled for that, the same body keeps its operands
// in registers where the wide form spills (840 x 1440, float64 inverse with the frequency sum: 272 -> 215 us per 8 scales).
template <typename T, int MODE, int SPEC>
__global__ void __launch_bounds__((MODE == 0 && SPEC == 3 && sizeof(T) == 8) ? 512 : NF_FFT_MAXTHREADS) k_fft_cols(nf::LineArgs<T> a) {
    fft_cols_body<T, MODE, SPEC>(a);
}

template <typename T, int MODE, int SPEC>
__global__ void __launch_bounds__(256) k_fft_cols_s(nf::LineArgs<T> a) {
    fft_cols_body<T, MODE, SPEC>(a);
}

// ------------------------------------------------------------------------------------------------------
// 4096-long columns in TWO passes of 64-point transforms (4096 = 64 x 64) over tiles of 32 adjacent columns.
// A 4096-long column is 64 KB (fp64): the single-pass kernel fits two of them in LDS - 32-byte runs per row, one workgroup per
// CU whose load / butterfly / store phases overlap nothing (2.3-2.9 TB/s).  Here a workgroup holds 64 rows x 32 columns
// (512-byte runs, 32 KB of LDS, 256 threads: five workgroups per CU) and each pass streams at nearly the rate of a copy
// (tools/col2pass.cpp); the plane is read and written twice, which still beats the one slow pass when the load carries the
// symmetrised filter (the matched filter's data inverse).
//   n = 64 a + r1, k = k1 + 64 k2:   Y[r1][k1] = w4096^(r1 k1) * sum_a x[64 a + r1] w64^(a k1)     (first pass, one workgroup per r1)
//                                    X[k1 + 64 k2] = sum_r1 Y[r1][k1] w64^(r1 k2)                  (second pass, one per k1)
// The first pass reads rows r1 + 64 a of `in` through the SPEC 3 load of k_fft_cols (filter, plane factor, window, packed
// column 0 / W/2) and writes Y[r1][k1] to row r1 + 64 k1 of `out`; the second pass reads rows 64 k1 + r1 and writes X[k1 + 64 k2]
// IN PLACE to row 64 k1 + k2: transformed row y lies at row ((y & 63) << 6) | (y >> 6), which the row pass that follows is told
// (LineArgs::rowSwap64) - it reads whole rows, in any order at the same cost.  A thread: radix-8 butterfly, twiddle, one LDS
// exchange, radix-8 butterfly.  tw[i] = exp(-2 pi i * i / 4096).
#define NF_C2P_CT 32
template <typename T>
__device__ __forceinline__ void c2p_core(cpx<T>* __restrict__ lds, cpx<T>* x, const cpx<T>* __restrict__ tw, int c, int j) {
    Dft<T, 8>::run(x, nullptr);
#pragma unroll
    for (int p = 1; p < 8; ++p) x[p] = cmul(x[p], tw[64 * j * p]);
#pragma unroll
    for (int p = 0; p < 8; ++p) lds[(j * 8 + p) * NF_C2P_CT + c] = x[p];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) x[jj] = lds[(jj * 8 + j) * NF_C2P_CT + c];
    Dft<T, 8>::run(x, nullptr);
}

template <typename T>
__global__ void __launch_bounds__(NF_C2P_CT * 8) k_cols2p_first(nf::LineArgs<T> a, const cpx<T>* __restrict__ tw) {
    __shared__ cpx<T> lds[64 * NF_C2P_CT];
    // (tile 0, whose first lane has the packed column's extra fetches, is dealt out first; then tiles side by side, row group after row group)
    int tile = 0, r1 = (int)blockIdx.x;
    if (blockIdx.x >= 64u) {
        const unsigned rest = blockIdx.x - 64u, nt = (unsigned)a.nTiles - 1u;
        tile = 1 + (int)(rest % nt);
        r1 = (int)(rest / nt);
    }
    const int c = threadIdx.x & (NF_C2P_CT - 1), j = threadIdx.x / NF_C2P_CT;
    const bool ok = tile * NF_C2P_CT + c < a.nLines;
    // (lanes beyond the last column repeat its loads and store nothing: no branch around the loads, which stay one batch)
    const int col = ok ? tile * NF_C2P_CT + c : a.nLines - 1;
    Geo G;
    G.inP = (const char*)a.in + (size_t)blockIdx.y * (size_t)a.inBatchStride * sizeof(cpx<T>);
    cpx<T>* __restrict__ outP = (cpx<T>*)a.out + (size_t)blockIdx.y * (size_t)a.outBatchStride;
    cpx<T> x[8];
    unsigned off[8];
    // every operand of the load in ONE batch of independent fetches (data, symmetrised filter, window), the arithmetic after them
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const unsigned e = (unsigned)(r1 + 64 * (j + 8 * m));
        off[m] = (unsigned)col * (unsigned)a.inLineStride + e * (unsigned)a.inElemStride;
        x[m] = ((const cpx<T>*)G.inP)[off[m]];
    }
    // fused_ops_herm for one half-stored plane (k_fft_cols, SPEC 3: the same order of roundings)
    double f[8];
    if (a.mulSym != nullptr) {
        const double* __restrict__ ms = a.mulSym + (size_t)blockIdx.y * (size_t)a.mulBatchStride;
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = ms[off[m]];
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = 1.0;
    }
    double wy[8], wl = 1.0;
    const bool win = a.winElem != nullptr;
    if (win) {
        wl = a.winLine[col];
#pragma unroll
        for (int m = 0; m < 8; ++m) wy[m] = a.winElem[r1 + 64 * (j + 8 * m)];
    }
    if (a.usePlaneScale) {
        const double ps = a.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)];
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] *= ps;
    }
    if (win) {
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] *= wy[m] * wl;
    }
    const T sgn = (T)a.conjInSign;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        x[m].x = (T)((double)x[m].x * f[m]);
        x[m].y = (T)((double)(x[m].y * sgn) * f[m]);
    }
    if (a.packNyq && col == 0) {
        // columns 0 and W/2 as ONE complex column z0 + i zN (see k_fft_cols): this one lane of tile 0 fetches both columns again
        // and applies fused_ops_herm's own order of roundings, (value x filter x plane factor) x window - four elements per batch
        // of loads (one lane's sequential fetches would otherwise be the tail of the whole launch)
        const unsigned colN = (unsigned)(a.realW / 2);
        const double* __restrict__ ms = a.mulSym ? a.mulSym + (size_t)blockIdx.y * (size_t)a.mulBatchStride : nullptr;
        const double ps = a.usePlaneScale ? a.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)] : 1.0;
        const double wlN = win ? a.winLine[colN] : 1.0;
#pragma unroll
        for (int m0 = 0; m0 < 8; m0 += 4) {
            cpx<T> z0[4], zN[4];
            double g0[4], gN[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned offN = off[m0 + u] + colN * (unsigned)a.inLineStride;
                z0[u] = ((const cpx<T>*)G.inP)[off[m0 + u]];
                zN[u] = ((const cpx<T>*)G.inP)[offN];
                g0[u] = ms ? ms[off[m0 + u]] : 1.0;
                gN[u] = ms ? ms[offN] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (a.usePlaneScale) {
                    g0[u] *= ps;
                    gN[u] *= ps;
                }
                double re0 = (double)z0[u].x * g0[u], im0 = (double)(z0[u].y * sgn) * g0[u];
                double reN = (double)zN[u].x * gN[u], imN = (double)(zN[u].y * sgn) * gN[u];
                if (win) {
                    const double w0 = wy[m0 + u] * wl, wN = wy[m0 + u] * wlN;
                    re0 *= w0;
                    im0 *= w0;
                    reN *= wN;
                    imN *= wN;
                }
                x[m0 + u].x = (T)re0 - (T)imN;
                x[m0 + u].y = (T)im0 + (T)reN;
            }
        }
    }
    c2p_core<T>(lds, x, tw, c, j);
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k1 = j + 8 * q;
        if (r1 * k1 != 0) x[q] = cmul(x[q], tw[r1 * k1]);
        outP[(unsigned)(r1 + 64 * k1) * (unsigned)a.outElemStride + (unsigned)col * (unsigned)a.outLineStride] = x[q];
    }
}

template <typename T>
__global__ void __launch_bounds__(NF_C2P_CT * 8) k_cols2p_second(nf::LineArgs<T> a, const cpx<T>* __restrict__ tw) {
    __shared__ cpx<T> lds[64 * NF_C2P_CT];
    const int tile = (int)(blockIdx.x % (unsigned)a.nTiles), k1 = (int)(blockIdx.x / (unsigned)a.nTiles);
    const int c = threadIdx.x & (NF_C2P_CT - 1), j = threadIdx.x / NF_C2P_CT;
    const bool ok = tile * NF_C2P_CT + c < a.nLines;
    const int col = ok ? tile * NF_C2P_CT + c : a.nLines - 1;
    cpx<T>* __restrict__ outP = (cpx<T>*)a.out + (size_t)blockIdx.y * (size_t)a.outBatchStride;
    cpx<T> x[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        x[m] = outP[(unsigned)(64 * k1 + j + 8 * m) * (unsigned)a.outElemStride + (unsigned)col * (unsigned)a.outLineStride];
    c2p_core<T>(lds, x, tw, c, j);
    if (!ok) return;
    const T r = (T)a.scale, ri = r * (T)a.conjOutSign;
    if (a.packNyq && col == 0) {
        // inverse of the packed column: the two real columns come back as real and imaginary part
        const unsigned colN = (unsigned)(a.realW / 2);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            cpx<T> v0, vN;
            v0.x = x[q].x * r;
            v0.y = (T)0;
            vN.x = x[q].y * ri;
            vN.y = (T)0;
            cpx<T>* orow = outP + (unsigned)(64 * k1 + j + 8 * q) * (unsigned)a.outElemStride;
            orow[0] = v0;
            orow[colN * (unsigned)a.outLineStride] = vN;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            cpx<T> v = x[q];
            v.x *= r;
            v.y *= ri;
            outP[(unsigned)(64 * k1 + j + 8 * q) * (unsigned)a.outElemStride + (unsigned)col * (unsigned)a.outLineStride] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The template path of a matched-filter scale with one frequency map, as ONE pass over the column tiles (float32 templates):
//   forward column transform of the template's row spectra (k_fft_cols, SPEC 0)      - result stays in LDS
//   filter build for the tile's modes and their mirror modes (k_build_filter_sym, N = 1) - filter cube (optional) and
//                                                                                       symmetrised filter go to HBM
//   spectrum x symmetrised filter x planeScale x pixel window, conjugated (the load of k_fft_cols, SPEC 3)
//   inverse column transform, stored for the rows that are read (k_fft_cols, SPEC 3)
// Per element the arithmetic of those three kernels in their order, so the results are theirs bit for bit; what is saved is the
// traffic between them: the template's spectrum (written once, read twice) and the symmetrised filter's read-back.
// After the forward DIT stages the tile lies in natural order; the inverse DIT stages want element k at the digit-reversed
// position rev[k]: every thread carries its elements' products over a barrier in registers (NF_TF_MAXE).
template <typename CT>
__device__ __forceinline__ float tf_filter_value(CT c, double rhs) {
    if (c == (CT)0) return 0.0f;
    const CT inv = (CT)1 / c;
    return (float)((double)inv * rhs);
}

// ---- the template path of a one-frequency scale with 4096-long columns in THREE passes (round 3, after k_cols2p_*) ----
// k_template_cols does the whole path in one pass but as one 128 KB workgroup per CU whose pointwise phase waits for memory
// latency (213 us per scale at 4096^2).  With the column transforms split 64 x 64 the same work is three streaming passes:
//   k_template2p_first   row spectra -> Y[r1][k1] (first pass of the forward transform), into `work`
//   k_template2p_middle  one workgroup per (column tile, k1): second pass of the forward transform - the workgroup then holds the
//                        spectrum at the modes k = k1 + 64 k2 of its columns -, filter and mirror-mode filter, symmetrised filter
//                        (and the cube where wanted) to HBM, product with plane factor and window, and the FIRST pass of the
//                        inverse transform over k2 (the layout a thread holds after one 64-point transform is the layout the next
//                        one starts from), twiddled, in place
//   k_template2p_last    second pass of the inverse transform, only the wanted rows stored (natural order, in place)
// The packed column 0 / W/2 needs Z[k] and Z[-k] together: the workgroups of tile 0 also transform column 0 of the mirror block
// k1' = (64 - k1) mod 64.  Per element the arithmetic of k_template_cols' pointwise phase; the transforms round differently
// (another factorisation), within float32 round-off.
__device__ __forceinline__ void c2p_block(unsigned bx, int nTiles, int& tile, int& blk) {
    // tile 0 (extra work for the packed column) is dealt out first; then tiles side by side, row group after row group
    tile = 0;
    blk = (int)bx;
    if (bx >= 64u) {
        const unsigned rest = bx - 64u, nt = (unsigned)nTiles - 1u;
        tile = 1 + (int)(rest % nt);
        blk = (int)(rest / nt);
    }
}

__global__ void __launch_bounds__(NF_C2P_CT * 8) k_template2p_first(nf::LineArgs<float> af, cpx<float>* __restrict__ work, long long workBatchStride,
                                                                    const cpx<float>* __restrict__ tw) {
    __shared__ cpx<float> lds[64 * NF_C2P_CT];
    int tile, r1;
    c2p_block(blockIdx.x, af.nTiles, tile, r1);
    const int c = threadIdx.x & (NF_C2P_CT - 1), j = threadIdx.x / NF_C2P_CT;
    const bool ok = tile * NF_C2P_CT + c < af.nLines;
    const int col = ok ? tile * NF_C2P_CT + c : af.nLines - 1;
    const cpx<float>* __restrict__ inP = (const cpx<float>*)af.in + (size_t)blockIdx.y * (size_t)af.inBatchStride;
    cpx<float>* __restrict__ outP = work + (size_t)blockIdx.y * (size_t)workBatchStride;
    cpx<float> x[8];
    unsigned off[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        off[m] = (unsigned)col * (unsigned)af.inLineStride + (unsigned)(r1 + 64 * (j + 8 * m)) * (unsigned)af.inElemStride;
        x[m] = inP[off[m]];
    }
    const float sgn = (float)af.conjInSign;
    if (af.packNyq && col == 0) {
        // forward: columns 0 and W/2 of the row spectra are real, z = c0 + i cN
        const unsigned colN = (unsigned)(af.realW / 2);
        cpx<float> zN[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) zN[m] = inP[off[m] + colN * (unsigned)af.inLineStride];
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m].y = zN[m].x;
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m].y *= sgn;
    }
    c2p_core<float>(lds, x, tw, c, j);
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k1 = j + 8 * q;
        if (r1 * k1 != 0) x[q] = cmul(x[q], tw[r1 * k1]);
        outP[(unsigned)(r1 + 64 * k1) * (unsigned)af.realW + (unsigned)col] = x[q];
        // (a second copy of the packed column in column W/2, which no pass reads as data: the middle pass works in place, and the
        //  workgroup of block k1 needs column 0 of block 64 - k1 as THIS pass left it)
        if (af.packNyq && col == 0) outP[(unsigned)(r1 + 64 * k1) * (unsigned)af.realW + (unsigned)(af.realW / 2)] = x[q];
    }
}

template <typename CT>
__global__ void __launch_bounds__(NF_C2P_CT * 8) k_template2p_middle(nf::LineArgs<float> af, nf::LineArgs<float> ai, nf::TemplateFuse tf,
                                                                     cpx<float>* __restrict__ work, long long workBatchStride,
                                                                     const cpx<float>* __restrict__ tw) {
    __shared__ cpx<float> lds[64 * NF_C2P_CT];
    __shared__ cpx<float> zmir[64];
    int tile, k1;
    c2p_block(blockIdx.x, af.nTiles, tile, k1);
    const int c = threadIdx.x & (NF_C2P_CT - 1), j = threadIdx.x / NF_C2P_CT;
    const bool ok = tile * NF_C2P_CT + c < af.nLines;
    const unsigned col = (unsigned)(ok ? tile * NF_C2P_CT + c : af.nLines - 1);
    const unsigned W = (unsigned)af.realW, colN = W / 2, n = 4096u;
    cpx<float>* __restrict__ wp = work + (size_t)blockIdx.y * (size_t)workBatchStride;
    // the noise power at the thread's modes and mirror modes, the window factors and the Y values: one batch of fetches
    const CT* __restrict__ cov = (const CT*)tf.cov;
    const unsigned xm = col == 0u ? 0u : W - col;
    const bool win = ai.winElem != nullptr;
    cpx<float> x[8];
    CT c1[8], c2[8];
    double wy[8], wl = 1.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) x[m] = wp[(unsigned)(64 * k1 + j + 8 * m) * W + col];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned e = (unsigned)(k1 + 64 * (j + 8 * q)), em = e == 0u ? 0u : n - e;
        c1[q] = cov[(size_t)e * W + col];
        c2[q] = cov[(size_t)em * W + xm];
    }
    if (win) {
        wl = ai.winLine[col];
#pragma unroll
        for (int q = 0; q < 8; ++q) wy[q] = ai.winElem[k1 + 64 * (j + 8 * q)];
    }
    c2p_core<float>(lds, x, tw, c, j);      // x[q] = Z[k1 + 64 (j + 8 q)] of this column
    const bool packed0 = af.packNyq && tile == 0;
    if (packed0) {
        // column 0 of the mirror block k1' = (64 - k1) mod 64: Z[-k] of the packed column
        const int k1m = (64 - k1) & 63;
        cpx<float> y[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            y[m].x = 0.0f;
            y[m].y = 0.0f;
            if (c == 0) y[m] = wp[(unsigned)(64 * k1m + j + 8 * m) * W + colN];      // (the first pass's copy of column 0)
        }
        __syncthreads();
        c2p_core<float>(lds, y, tw, c, j);
        if (c == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) zmir[j + 8 * q] = y[q];
        }
        __syncthreads();
    }
    const float r = (float)af.scale, ri = r * (float)af.conjOutSign, h = 0.5f;
    const float cj = (float)ai.conjInSign;
    const size_t npix = (size_t)af.realH * (size_t)W;
    double* __restrict__ sym = tf.sym + (size_t)blockIdx.y * npix;
    float* __restrict__ filt = tf.filt[blockIdx.y & (NF_TF_MAXBATCH - 1)];
    const double ps = ai.usePlaneScale ? ai.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)] : 1.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned k2 = (unsigned)(j + 8 * q), e = (unsigned)k1 + 64u * k2, em = e == 0u ? 0u : n - e;
        const cpx<float> z = x[q];
        if (packed0 && col == 0u) {
            // columns 0 and W/2 as they leave the forward pass (k_fft_cols' unpacking), each with its mirror row in its own column
            const cpx<float> zm = zmir[k1 != 0 ? 63u - k2 : ((64u - k2) & 63u)];
            cpx<float> g0, gN, g0m, gNm;
            g0.x = h * (z.x + zm.x) * r;
            g0.y = h * (z.y - zm.y) * ri;
            gN.x = h * (z.y + zm.y) * r;
            gN.y = -h * (z.x - zm.x) * ri;
            g0m.x = h * (zm.x + z.x) * r;
            g0m.y = h * (zm.y - z.y) * ri;
            gNm.x = h * (zm.y + z.y) * r;
            gNm.y = -h * (zm.x - z.x) * ri;
            const size_t p0 = (size_t)e * W, p0m = (size_t)em * W, pN = p0 + colN, pNm = p0m + colN;
            const float a1 = tf_filter_value<CT>(c1[q], tf.w0 * (double)hypotf(g0.x, g0.y));
            const float a2 = p0m == p0 ? a1 : tf_filter_value<CT>(c2[q], tf.w0 * (double)hypotf(g0m.x, g0m.y));
            const float b1 = tf_filter_value<CT>(cov[pN], tf.w0 * (double)hypotf(gN.x, gN.y));
            const float b2 = pNm == pN ? b1 : tf_filter_value<CT>(cov[pNm], tf.w0 * (double)hypotf(gNm.x, gNm.y));
            const double s0 = 0.5 * ((double)a1 + (double)a2), sN = 0.5 * ((double)b1 + (double)b2);
            if (filt != nullptr) {
                filt[p0] = a1;
                filt[pN] = b1;
            }
            sym[p0] = s0;
            sym[pN] = sN;
            double f0 = s0, fN = sN;
            if (ai.usePlaneScale) {
                f0 *= ps;
                fN *= ps;
            }
            double re0 = 0.0, im0 = 0.0, reN = 0.0, imN = 0.0;
            re0 += (double)g0.x * f0;
            im0 += (double)(g0.y * cj) * f0;
            reN += (double)gN.x * fN;
            imN += (double)(gN.y * cj) * fN;
            if (win) {
                const double w0_ = wy[q] * wl, wN_ = wy[q] * ai.winLine[colN];
                re0 *= w0_;
                im0 *= w0_;
                reN *= wN_;
                imN *= wN_;
            }
            const float x0 = (float)re0, y0 = (float)im0, xN = (float)reN, yN = (float)imN;
            x[q].x = x0 - yN;
            x[q].y = y0 + xN;
        } else {
            cpx<float> v = z;
            v.x *= r;
            v.y *= ri;                                        // the spectrum value as the forward pass would store it
            const size_t pp = (size_t)e * W + col, pm = (size_t)em * W + xm;
            const double rhs = tf.w0 * (double)hypotf(v.x, v.y);
            const float o1 = tf_filter_value<CT>(c1[q], rhs);
            const float o2 = pm != pp ? tf_filter_value<CT>(c2[q], rhs) : o1;
            const double sv = 0.5 * ((double)o1 + (double)o2);
            if (ok) {
                if (filt != nullptr) {
                    filt[pp] = o1;
                    if (pm != pp) filt[pm] = o2;
                }
                sym[pp] = sv;
            }
            double f = sv;
            if (ai.usePlaneScale) f *= ps;
            if (win) f *= wy[q] * wl;
            x[q].x = (float)((double)v.x * f);
            x[q].y = (float)((double)(v.y * cj) * f);
        }
    }
    // first pass of the inverse transform over k2 (index a = j + 8 q, which is where the values are), twiddled by w^(k1 o1)
    __syncthreads();
    c2p_core<float>(lds, x, tw, c, j);
    if (!ok) return;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int o1 = j + 8 * q;
        if (k1 * o1 != 0) x[q] = cmul(x[q], tw[k1 * o1]);
        wp[(unsigned)(64 * k1 + o1) * W + col] = x[q];
    }
}

__global__ void __launch_bounds__(NF_C2P_CT * 8) k_template2p_last(nf::LineArgs<float> ai, cpx<float>* __restrict__ work, long long workBatchStride,
                                                                   const cpx<float>* __restrict__ tw) {
    __shared__ cpx<float> lds[64 * NF_C2P_CT];
    const int tile = (int)(blockIdx.x % (unsigned)ai.nTiles), o1 = (int)(blockIdx.x / (unsigned)ai.nTiles);
    const int c = threadIdx.x & (NF_C2P_CT - 1), j = threadIdx.x / NF_C2P_CT;
    const bool ok = tile * NF_C2P_CT + c < ai.nLines;
    const unsigned col = (unsigned)(ok ? tile * NF_C2P_CT + c : ai.nLines - 1);
    const unsigned W = (unsigned)ai.realW;
    cpx<float>* __restrict__ wp = work + (size_t)blockIdx.y * (size_t)workBatchStride;
    cpx<float> x[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) x[m] = wp[(unsigned)(o1 + 64 * (j + 8 * m)) * W + col];
    c2p_core<float>(lds, x, tw, c, j);
    if (!ok) return;
    const float r = (float)ai.scale, ri = r * (float)ai.conjOutSign;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = o1 + 64 * (j + 8 * q);
        if (ai.storeHi != 0 && (e < ai.storeLo || e >= ai.storeHi)) continue;
        cpx<float>* orow = wp + (unsigned)e * W;
        if (ai.packNyq && col == 0u) {
            cpx<float> v0, vN;
            v0.x = x[q].x * r;
            v0.y = 0.0f;
            vN.x = x[q].y * ri;
            vN.y = 0.0f;
            orow[0] = v0;
            orow[W / 2] = vN;
        } else {
            cpx<float> v = x[q];
            v.x *= r;
            v.y *= ri;
            orow[col] = v;
        }
    }
}

template <typename CT, int MODE>
__global__ void __launch_bounds__(NF_FFT_MAXTHREADS) k_template_cols(nf::LineArgs<float> af, nf::LineArgs<float> ai, nf::TemplateFuse tf) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<float>* buf = (cpx<float>*)nf_smem;
    const nf::LinePlan<float>& p = af.plan;
    const int per = (af.nTiles + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= af.nTiles) return;
    Geo G;
    G.TL = af.linesPerWG;
    G.line0 = tile * G.TL;
    const int nvalid = min(G.TL, af.nLines - G.line0);
    G.tl = threadIdx.x & (G.TL - 1);
    G.tb = threadIdx.x / G.TL;
    G.TB = blockDim.x / G.TL;
    G.nl = G.tl < nvalid ? G.tl + 1 : 0;
    G.inP = (const char*)af.in + (size_t)blockIdx.y * (size_t)af.inBatchStride * sizeof(cpx<float>);
    G.outP = (char*)ai.out + (size_t)blockIdx.y * (size_t)ai.outBatchStride * sizeof(cpx<float>);
    const bool packed0 = af.packNyq && tile == 0;
    bool pairF = false, pairI = false;
    if constexpr (MODE == 0) {
        const bool whole = G.TL >= 2 && nvalid == G.TL && !packed0 && p.nst >= 2 && (blockDim.x % (G.TL >> 1)) == 0;
        pairF = af.colPair && whole && ((size_t)G.inP & 15) == 0;
        pairI = ai.colPair && whole && ((size_t)G.outP & 15) == 0;
    }
    // ---- forward transform: every stage leaves its result in LDS ----
    for (int s = 0; s < p.nst; ++s) {
        bool done = false;
        if constexpr (MODE == 0) {
            if (pairF && s == 0) done = fft_stage_colpair_dispatch<0>(buf, af, s, true, G);
        }
        if (!done) fft_stage_dispatch<float, true, MODE, 0>(buf, af, s, false, s == 0, false, G);
    }
    // ---- the tile's modes: filter, symmetrised filter, product ----
    const unsigned n = (unsigned)p.n, TL = (unsigned)G.TL, W = (unsigned)af.realW, colN = W / 2;
    const float r = (float)af.scale, ri = r * (float)af.conjOutSign, h = 0.5f;
    const float cj = (float)ai.conjInSign;
    const CT* __restrict__ cov = (const CT*)tf.cov;
    const size_t npix = (size_t)af.realH * (size_t)W;
    double* __restrict__ sym = tf.sym + (size_t)blockIdx.y * npix;
    float* __restrict__ filt = tf.filt[blockIdx.y & (NF_TF_MAXBATCH - 1)];
    const double ps = ai.usePlaneScale ? ai.planeScale[blockIdx.y & (NF_PLANE_SCALES - 1)] : 1.0;
    const bool win = ai.winElem != nullptr;
    cpx<float> hold[NF_TF_MAXE];
    // (a thread's elements in groups of NF_TF_GRP: the noise power at the group's modes and mirror modes is fetched in one
    //  batch of independent loads - one memory round trip per group instead of one per element)
#define NF_TF_GRP 8
#pragma unroll
    for (int j0 = 0; j0 < NF_TF_MAXE; j0 += NF_TF_GRP) {
        CT c1[NF_TF_GRP], c2[NF_TF_GRP];
#pragma unroll
        for (int u = 0; u < NF_TF_GRP; ++u) {
            const unsigned i = threadIdx.x + (unsigned)(j0 + u) * blockDim.x;
            const unsigned line = i & (TL - 1), e = i / TL;
            const bool live = i < n * TL && (int)line < nvalid && !(packed0 && line == 0);
            const unsigned em = e == 0u ? 0u : n - e;
            const unsigned col = (unsigned)G.line0 + line;
            const unsigned xm = col == 0u ? 0u : W - col;
            c1[u] = live ? cov[(size_t)e * W + col] : (CT)0;
            c2[u] = live ? cov[(size_t)em * W + xm] : (CT)0;
        }
#pragma unroll
        for (int u = 0; u < NF_TF_GRP; ++u) {
            const int j = j0 + u;
            const unsigned i = threadIdx.x + (unsigned)j * blockDim.x;
            hold[j].x = 0.0f;
            hold[j].y = 0.0f;
            if (i >= n * TL) continue;
            const unsigned line = i & (TL - 1), e = i / TL;
            if ((int)line >= nvalid) continue;
            const unsigned em = e == 0u ? 0u : n - e;
            const cpx<float> z = buf[(unsigned)swz((int)e) * TL + line];
            if (packed0 && line == 0) {
                // columns 0 and W/2 as they leave the forward pass (k_fft_cols' unpacking), each with its mirror row in its own column
                const cpx<float> zm = buf[(unsigned)swz((int)em) * TL];
                cpx<float> g0, gN, g0m, gNm;
                g0.x = h * (z.x + zm.x) * r;
                g0.y = h * (z.y - zm.y) * ri;
                gN.x = h * (z.y + zm.y) * r;
                gN.y = -h * (z.x - zm.x) * ri;
                g0m.x = h * (zm.x + z.x) * r;
                g0m.y = h * (zm.y - z.y) * ri;
                gNm.x = h * (zm.y + z.y) * r;
                gNm.y = -h * (zm.x - z.x) * ri;
                const size_t p0 = (size_t)e * W, p0m = (size_t)em * W, pN = p0 + colN, pNm = p0m + colN;
                const float a1 = tf_filter_value<CT>(cov[p0], tf.w0 * (double)hypotf(g0.x, g0.y));
                const float a2 = p0m == p0 ? a1 : tf_filter_value<CT>(cov[p0m], tf.w0 * (double)hypotf(g0m.x, g0m.y));
                const float b1 = tf_filter_value<CT>(cov[pN], tf.w0 * (double)hypotf(gN.x, gN.y));
                const float b2 = pNm == pN ? b1 : tf_filter_value<CT>(cov[pNm], tf.w0 * (double)hypotf(gNm.x, gNm.y));
                const double s0 = 0.5 * ((double)a1 + (double)a2), sN = 0.5 * ((double)b1 + (double)b2);
                if (filt != nullptr) {
                    filt[p0] = a1;
                    filt[pN] = b1;
                }
                sym[p0] = s0;
                sym[pN] = sN;
                // the inverse pass's load of the packed column (fused_ops_herm on each column, then z0 + i zN)
                double f0 = s0, fN = sN;
                if (ai.usePlaneScale) {
                    f0 *= ps;
                    fN *= ps;
                }
                double re0 = 0.0, im0 = 0.0, reN = 0.0, imN = 0.0;
                re0 += (double)g0.x * f0;
                im0 += (double)(g0.y * cj) * f0;
                reN += (double)gN.x * fN;
                imN += (double)(gN.y * cj) * fN;
                if (win) {
                    const double w0_ = ai.winElem[e] * ai.winLine[0], wN_ = ai.winElem[e] * ai.winLine[colN];
                    re0 *= w0_;
                    im0 *= w0_;
                    reN *= wN_;
                    imN *= wN_;
                }
                const float x0 = (float)re0, y0 = (float)im0, xN = (float)reN, yN = (float)imN;
                hold[j].x = x0 - yN;
                hold[j].y = y0 + xN;
            } else {
                cpx<float> v = z;
                v.x *= r;
                v.y *= ri;                                        // the spectrum value as the forward pass would store it
                const unsigned col = (unsigned)G.line0 + line;
                const unsigned xm = col == 0u ? 0u : W - col;
                const bool selfColumn = xm == col;
                const size_t pp = (size_t)e * W + col, pm = (size_t)em * W + xm;
                const float o1 = tf_filter_value<CT>(c1[u], tf.w0 * (double)hypotf(v.x, v.y));
                float o2 = o1;
                if (pm != pp) {
                    double rhs2 = tf.w0 * (double)hypotf(v.x, v.y);
                    if (selfColumn) {                             // the mirror mode is stored in this column: its own spectrum value
                        cpx<float> vm = buf[(unsigned)swz((int)em) * TL + line];
                        vm.x *= r;
                        vm.y *= ri;
                        rhs2 = tf.w0 * (double)hypotf(vm.x, vm.y);
                    }
                    o2 = tf_filter_value<CT>(c2[u], rhs2);
                }
                const double sv = 0.5 * ((double)o1 + (double)o2);
                if (filt != nullptr) {
                    filt[pp] = o1;
                    if (!selfColumn) filt[pm] = o2;
                }
                sym[pp] = sv;
                // the inverse pass's load (SPEC 3): one real factor, formed in its order
                double f = sv;
                if (ai.usePlaneScale) f *= ps;
                if (win) f *= ai.winElem[e] * ai.winLine[col];
                hold[j].x = (float)((double)v.x * f);
                hold[j].y = (float)((double)(v.y * cj) * f);
            }
        }
    }
#undef NF_TF_GRP
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NF_TF_MAXE; ++j) {
        const unsigned i = threadIdx.x + (unsigned)j * blockDim.x;
        if (i >= n * TL) continue;
        const unsigned line = i & (TL - 1), e = i / TL;
        if ((int)line >= nvalid) continue;
        buf[(unsigned)swz((int)p.rev[e]) * TL + line] = hold[j];
    }
    __syncthreads();
    // ---- inverse transform from LDS; the last stage stores the rows that are wanted ----
    for (int s = 0; s < p.nst; ++s) {
        const bool dstG = s == p.nst - 1;
        bool done = false;
        if constexpr (MODE == 0) {
            if (pairI && dstG) done = fft_stage_colpair_dispatch<3>(buf, ai, s, false, G);
        }
        if (!done) fft_stage_dispatch<float, true, MODE, 3>(buf, ai, s, false, false, dstG, G);
    }
}

// ------------------------------------------------------------------------------------------------------
// Circular filtering of REAL lines by a real, even transfer function: y = IFFT(FFT(x) * w), one kernel, one pass over the map.
// enmap.apply_window (nemo/filters.py:98-101, nemo/maps.py:1849) multiplies the 2-D spectrum by wy[ky] * wx[kx]: the two 1-D
// operators along x and along y commute, each is this kernel once - the map is read and written twice instead of four times,
// and no spectrum ever reaches HBM.  Two real lines travel as one complex line z = a + i b; a real, even w acts on both alike
// and the inverse transform hands them back as real and imaginary part.  In LDS: natural order -> DIF stages -> multiply by
// w at the digit-reversed positions (table wperm[rev[k]] = w[k] / n, given by the caller) and conjugate -> DIT stages ->
// conjugate at the store.  Plain-radix lengths only (MODE 0 / 1).
//   rows (COLS false): line p = rows 2p, 2p+1 of the map;  columns (COLS true): lane c of a tile = columns 2c, 2c+1.
// a.winElem = wperm; a.inType / a.outType 1 = float64, 2 = float32; a.epiZeroRef: where that map (of the output type) is zero
// the output is zero (nemo/filters.py:99-101).
template <typename T>
__device__ __forceinline__ T lf_load(const void* p, int type, size_t off) {
    return type == 1 ? (T)((const double*)p)[off] : (T)((const float*)p)[off];
}

template <typename T>
__device__ __forceinline__ void lf_store(const nf::LineArgs<T>& a, void* outP, const void* refP, size_t off, T v, bool useRef) {
    if (a.outType == 1) {
        double w = (double)v;
        if (useRef && ((const double*)refP)[off] == 0.0) w = 0.0;
        ((double*)outP)[off] = w;
    } else {
        float w = (float)v;
        if (useRef && ((const float*)refP)[off] == 0.0f) w = 0.0f;
        ((float*)outP)[off] = w;
    }
}

template <typename T, bool COLS, int MODE>
__global__ void __launch_bounds__(COLS ? NF_FFT_MAXTHREADS : 512) k_line_filter(nf::LineArgs<T> a) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<T>* buf = (cpx<T>*)nf_smem;
    const nf::LinePlan<T>& p = a.plan;
    const int n = p.n;
    Geo G;
    int nvalid = 0;
    if (COLS) {
        const int per = (a.nTiles + 7) >> 3;
        const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (tile >= a.nTiles) return;
        G.TL = a.linesPerWG;
        G.line0 = tile * G.TL;
        nvalid = min(G.TL, a.nLines - G.line0);
        G.tl = threadIdx.x & (G.TL - 1);
        G.tb = threadIdx.x / G.TL;
        G.TB = blockDim.x / G.TL;
        G.nl = G.tl < nvalid ? G.tl + 1 : 0;
    } else {
        G.line0 = blockIdx.x * a.linesPerWG;
        G.nl = min(a.linesPerWG, a.nLines - G.line0);
        G.TB = a.tb;
        G.TL = blockDim.x / a.tb;
        G.tl = threadIdx.x / a.tb;
        G.tb = threadIdx.x - G.tl * a.tb;
    }
    const size_t plane = (size_t)a.realH * (size_t)a.realW;
    const size_t inEs = a.inType == 1 ? 8 : 4, outEs = a.outType == 1 ? 8 : 4;
    G.inP = (const char*)a.in + (size_t)blockIdx.y * plane * inEs;
    G.outP = (char*)a.out + (size_t)blockIdx.y * plane * outEs;
    // (every map of a batch has its own zero reference and its own row flags, laid out like the maps)
    const void* refP = a.epiZeroRef != nullptr ? (const void*)((const char*)a.epiZeroRef + (size_t)blockIdx.y * plane * outEs) : nullptr;
    const int* __restrict__ zeroRows = a.epiZeroRows != nullptr ? a.epiZeroRows + (size_t)blockIdx.y * (size_t)a.realH : nullptr;
    constexpr int NB = 4;
    const size_t W = (size_t)a.realW;

    // 1. two real lines -> one complex line, natural order
    if (COLS) {
        if (G.nl > 0) {
            const size_t c0 = 2u * (size_t)(G.line0 + G.tl);
            const bool haveB = c0 + 1 < W;
            // the lane's two columns as one 2-vector where the pitch and the pointer allow it
            const bool vecD = a.inType == 1 && (W & 1) == 0 && ((size_t)G.inP & 15) == 0;
            const bool vecF = a.inType != 1 && (W & 1) == 0 && ((size_t)G.inP & 7) == 0;
            for (int e0 = G.tb; e0 < n; e0 += NB * G.TB) {
                cpx<T> z[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = min(e0 + j * G.TB, n - 1);
                    if (vecD) {
                        const double2 v = *(const double2*)((const double*)G.inP + (size_t)e * W + c0);
                        z[j].x = (T)v.x;
                        z[j].y = (T)v.y;
                    } else if (vecF) {
                        const float2 v = *(const float2*)((const float*)G.inP + (size_t)e * W + c0);
                        z[j].x = (T)v.x;
                        z[j].y = (T)v.y;
                    } else {
                        z[j].x = lf_load<T>(G.inP, a.inType, (size_t)e * W + c0);
                        z[j].y = haveB ? lf_load<T>(G.inP, a.inType, (size_t)e * W + c0 + 1) : (T)0;
                    }
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = e0 + j * G.TB;
                    if (e < n) buf[(unsigned)swz(e) * (unsigned)G.TL + (unsigned)G.tl] = z[j];
                }
            }
        }
    } else {
        for (int line = G.tl; line < G.nl; line += G.TL) {
            const size_t rowA = 2u * (size_t)(G.line0 + line);
            const bool haveB = rowA + 1 < (size_t)a.realH;
            cpx<T>* __restrict__ lb = buf + line * p.mPad;
            for (int e0 = G.tb; e0 < n; e0 += NB * G.TB) {
                cpx<T> z[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = min(e0 + j * G.TB, n - 1);
                    z[j].x = lf_load<T>(G.inP, a.inType, rowA * W + (size_t)e);
                    z[j].y = haveB ? lf_load<T>(G.inP, a.inType, (rowA + 1) * W + (size_t)e) : (T)0;
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int e = e0 + j * G.TB;
                    if (e < n) lb[swz(e)] = z[j];
                }
            }
        }
    }
    __syncthreads();
    // 2. forward transform, decimation in frequency: the spectrum lands on the digit-reversed positions
    for (int s = p.nst - 1; s >= 0; --s) fft_stage_dispatch<T, COLS, MODE, 0>(buf, a, s, true, false, false, G);
    // 3. times the transfer function; conjugate (the way back is a forward transform of the conjugate)
    for (int e = G.tb; e < n; e += G.TB) {
        const T w = (T)a.winElem[e];
        for (int line = G.tl; line < G.nl; line += (COLS ? G.nl : G.TL)) {
            const unsigned at = COLS ? (unsigned)swz(e) * (unsigned)G.TL + (unsigned)line : (unsigned)line * (unsigned)p.mPad + (unsigned)swz(e);
            cpx<T> v = buf[at];
            v.x = v.x * w;
            v.y = -(v.y * w);
            buf[at] = v;
        }
    }
    __syncthreads();
    // 4. back, decimation in time: digit-reversed -> natural
    for (int s = 0; s < p.nst; ++s) fft_stage_dispatch<T, COLS, MODE, 0>(buf, a, s, false, false, false, G);
    // 5. the two real lines: real part and minus the imaginary part
    if (COLS) {
        if (G.nl > 0) {
            const size_t c0 = 2u * (size_t)(G.line0 + G.tl);
            const bool haveB = c0 + 1 < W;
            const bool useRef = a.epiZeroRef != nullptr;
            const bool vecD = !useRef && a.outType == 1 && (W & 1) == 0 && ((size_t)G.outP & 15) == 0;
            const bool vecF = !useRef && a.outType != 1 && (W & 1) == 0 && ((size_t)G.outP & 7) == 0;
            for (int e = G.tb; e < n; e += G.TB) {
                const cpx<T> v = buf[(unsigned)swz(e) * (unsigned)G.TL + (unsigned)G.tl];
                if (vecD) {
                    double2 o;
                    o.x = (double)v.x;
                    o.y = (double)-v.y;
                    *(double2*)((double*)G.outP + (size_t)e * W + c0) = o;
                } else if (vecF) {
                    float2 o;
                    o.x = (float)v.x;
                    o.y = (float)-v.y;
                    *(float2*)((float*)G.outP + (size_t)e * W + c0) = o;
                } else {
                    lf_store<T>(a, G.outP, refP, (size_t)e * W + c0, v.x, useRef);
                    if (haveB) lf_store<T>(a, G.outP, refP, (size_t)e * W + c0 + 1, -v.y, useRef);
                }
            }
        }
    } else {
        for (int line = G.tl; line < G.nl; line += G.TL) {
            const size_t rowA = 2u * (size_t)(G.line0 + line);
            const bool haveB = rowA + 1 < (size_t)a.realH;
            const cpx<T>* __restrict__ lb = buf + line * p.mPad;
            // (a row the caller knows to hold no zero is not looked up in the zero reference)
            const bool refA = refP != nullptr && (zeroRows == nullptr || zeroRows[rowA] != 0);
            const bool refB = haveB && refP != nullptr && (zeroRows == nullptr || zeroRows[rowA + 1] != 0);
            for (int e = G.tb; e < n; e += G.TB) {
                const cpx<T> v = lb[swz(e)];
                lf_store<T>(a, G.outP, refP, rowA * W + (size_t)e, v.x, refA);
                if (haveB) lf_store<T>(a, G.outP, refP, (rowA + 1) * W + (size_t)e, -v.y, refB);
            }
        }
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

template <typename V>
static int upload_vec(const std::vector<V>& v, const V** dst) {
    void* d = nullptr;
    NF_HIP(hipMalloc(&d, v.size() * sizeof(V)));
    NF_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice));
    *dst = (const V*)d;
    return NF_OK;
}

template <typename T>
int make_line_plan(int n, LinePlan<T>* p, std::vector<unsigned short>* revOut = nullptr) {
    memset(p, 0, sizeof(*p));
    p->n = n;
    const long double pi = 3.141592653589793238462643383279502884L;
    int fac[NF_MAXFAC], nfac = 0;
    if (factorize(n, fac, &nfac)) {
        p->m = n;
        p->bluestein = 0;
    } else {
        p->bluestein = 1;
        p->m = next_smooth(2 * n - 1);
        if (!factorize(p->m, fac, &nfac)) {
            nf_set_error("internal: smooth length failed to factorize");
            return NF_ERR_SIZE;
        }
    }
    const int m = p->m;
    if (m >= 32768) {
        nf_set_error("FFT line too long for the single-pass LDS transform");
        return NF_ERR_SIZE;
    }
    p->mPad = (m + 31) & ~31;
    p->nst = nfac;
    // stage tables
    std::vector<cpx<T>> stw;
    int L = 1;
    for (int s = 0; s < nfac; ++s) {
        StageInfo& st = p->st[s];
        st.R = fac[s];
        st.L = L;
        st.magic = L > 1 ? (unsigned)((0x100000000ULL + (unsigned long long)L - 1ULL) / (unsigned long long)L) : 0u;
        st.twOff = (int)stw.size();
        if (L > 1) {
            const long long span = (long long)L * st.R;
            for (int k = 1; k < st.R; ++k)
                for (int q = 0; q < L; ++q) {
                    long double ang = -2.0L * pi * (long double)(((long long)q * k) % span) / (long double)span;
                    cpx<T> w;
                    w.x = (T)cosl(ang);
                    w.y = (T)sinl(ang);
                    stw.push_back(w);
                }
        }
        st.rootOff = (int)stw.size();
        if (st.R > 4 && st.R != 8) {
            for (int j = 0; j < st.R; ++j) {
                long double ang = -2.0L * pi * (long double)j / (long double)st.R;
                cpx<T> w;
                w.x = (T)cosl(ang);
                w.y = (T)sinl(ang);
                stw.push_back(w);
            }
        }
        L *= st.R;
    }
    if (stw.empty()) stw.resize(1);
    int rc = upload_vec<cpx<T>>(stw, &p->stw);
    if (rc) return rc;
    std::vector<unsigned short> rev(m), src(m);
    for (int i = 0; i < m; ++i) {
        int pos = host_digit_reverse(i, m, fac, nfac);
        rev[i] = (unsigned short)pos;
        src[pos] = (unsigned short)i;
    }
    if (revOut) *revOut = rev;
    rc = upload_vec<unsigned short>(rev, &p->rev);
    if (rc) return rc;
    rc = upload_vec<unsigned short>(src, &p->src);
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
            int pos = rev[f];
            cf[pos].x = (T)(bre[f] / (long double)m);
            cf[pos].y = (T)(bim[f] / (long double)m);
        }
        rc = upload_vec<cpx<T>>(chirp, &p->chirp);
        if (rc) return rc;
        rc = upload_vec<cpx<T>>(cf, &p->chirpFp);
        if (rc) return rc;
    }
    return NF_OK;
}

template <typename T>
void free_line_plan(LinePlan<T>* p) {
    if (p->stw) (void)hipFree((void*)p->stw);
    if (p->rev) (void)hipFree((void*)p->rev);
    if (p->src) (void)hipFree((void*)p->src);
    if (p->chirp) (void)hipFree((void*)p->chirp);
    if (p->chirpFp) (void)hipFree((void*)p->chirpFp);
    memset(p, 0, sizeof(*p));
}


static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

// A column pass over a small map has few tiles (840 x 1440: 180 tiles of four columns for 256 CUs): narrower tiles, down to one
// column, until there are two workgroups per CU - the short runs are merged by the L2 (neighbouring tiles run on the same XCD),
// an idle CU is not.  NF_FFT_COL_FILL=0 switches it off.
template <typename T>
static void fit_col_tiles(LineArgs<T>& a, int batch) {
    static const int fill = env_int("NF_FFT_COL_FILL", 512);
    while (a.linesPerWG > 1 && (long long)a.nTiles * batch < fill) {
        a.linesPerWG /= 2;
        a.nTiles = nf_div_up(a.nLines, a.linesPerWG);
    }
}

template <typename T>
int launch_lines(const LineArgs<T>& a0, bool cols, int batch, int threads, hipStream_t stream) {
    LineArgs<T> a = a0;
    if (cols) fit_col_tiles<T>(a, batch);
    size_t lds = (size_t)a.plan.mPad * a.linesPerWG * sizeof(cpx<T>);
    if (lds > (size_t)kMaxLds) {
        nf_set_error("FFT line of this length does not fit in LDS");
        return NF_ERR_SIZE;
    }
    if ((double)a.nLines * a.plan.n >= 2147483648.0) {
        nf_set_error("map plane too large (offsets inside a plane are 32-bit)");
        return NF_ERR_SIZE;
    }
    if (cols && a.inType != 0) {
        nf_set_error("internal: the column pass takes complex input");
        return NF_ERR_ARG;
    }
    static const int pairOn = env_int("NF_FFT_COL_PAIR", 1);
    // (long columns only - one workgroup per CU; with short ones the narrower tiles' extra workgroups are worth more: 840-long
    // columns 25.5 -> 28.3 us with pairs, 4096-long ones 81.9 -> 79.2 us)
    a.colPair = (cols && pairOn && sizeof(T) == 4 && lds > 80 * 1024 && a.outType == 0 && a.inLineStride == 1 && a.outLineStride == 1 &&
                 (a.inElemStride & 1) == 0 && (a.outElemStride & 1) == 0 && (a.inBatchStride & 1) == 0 && (a.outBatchStride & 1) == 0 &&
                 (a.linesPerWG & 1) == 0 && ((size_t)a.mulSym & 15) == 0 && (a.mulBatchStride & 1) == 0)
                    ? 1 : 0;
    static bool attrSet[2] = {false, false};
    int which = sizeof(T) == 8 ? 1 : 0;
    if (!attrSet[which]) {
#define NF_ATTR(K) NF_HIP(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds))
#define NF_ATTR3(KK, M) \
    NF_ATTR((KK<T, M, 0>)); \
    NF_ATTR((KK<T, M, 1>)); \
    NF_ATTR((KK<T, M, 2>))
        NF_ATTR3(k_fft_rows, 0);
        NF_ATTR3(k_fft_rows, 1);
        NF_ATTR3(k_fft_rows, 2);
        NF_ATTR3(k_fft_cols, 0);
        NF_ATTR3(k_fft_cols, 1);
        NF_ATTR3(k_fft_cols, 2);
        NF_ATTR((k_fft_cols<T, 0, 3>));
        NF_ATTR((k_fft_cols<T, 1, 3>));
        NF_ATTR((k_fft_cols<T, 2, 3>));
        NF_ATTR3(k_fft_cols_s, 0);
        NF_ATTR3(k_fft_cols_s, 1);
        NF_ATTR3(k_fft_cols_s, 2);
        NF_ATTR((k_fft_cols_s<T, 0, 3>));
        NF_ATTR((k_fft_cols_s<T, 1, 3>));
        NF_ATTR((k_fft_cols_s<T, 2, 3>));
#undef NF_ATTR3
#undef NF_ATTR
        attrSet[which] = true;
    }
    const double npixTotal = ((double)a.nLines + (cols && a.packNyq ? 1.0 : 0.0)) * a.plan.n * batch;
    const double inB = a.inType == 0 ? sizeof(cpx<T>) : (a.inType == 1 ? 8.0 : 4.0);
    const double outB = a.outType == 0 ? sizeof(cpx<T>) : (a.outType == 1 ? 8.0 : 4.0);
    const int tag = sizeof(T) == 8 ? (cols ? PT_FFT_COLS_F64 : PT_FFT_ROWS_F64) : (cols ? PT_FFT_COLS_F32 : PT_FFT_ROWS_F32);
    // algorithmic bytes per line element: pair lines carry two real rows (or two half-rows); the mirrored store writes the
    // element twice; the symmetrised load reads it (and its filter value) twice
    double lineIn = cols ? ((a.nSum > 1 ? a.nSum : 1) * (inB + (a.mulF32 ? 4.0 : 0.0) + (a.mulSym ? 8.0 : 0.0))) : inB, lineOut = outB;
    if (!cols && a.pairMode == 1) lineIn *= 2.0;
    if (!cols && a.pairMode == 2) lineOut *= 2.0;
    if (!cols && a.epiMask) lineIn += 2.0;
    if (!cols && a.epiZeroRef) lineIn += 2.0 * outB;
    if (cols && a.mirror) lineOut *= 2.0;
    if (cols && a.herm && !a.halfIn) lineIn *= 2.0;
    ProfScope ps(tag, stream, npixTotal * (lineIn + lineOut));
    int mode = a.plan.bluestein ? 2 : 0;
    for (int s = 0; s < a.plan.nst; ++s)
        if (a.plan.st[s].R > 8) mode = 1;
    if (a.plan.bluestein && mode == 1) {
        nf_set_error("internal: Bluestein length with a radix above 8");
        return NF_ERR_SIZE;
    }
    dim3 grid = cols ? dim3(((a.nTiles + 7) / 8) * 8, batch) : dim3(nf_div_up(a.nLines - a.lineBase, a.linesPerWG), batch);
    // 3: the fused load of ONE half-stored plane (batched form of 2)
    const int spec = cols ? (a.mirror ? 1 : (a.fused ? ((a.herm && a.halfIn && a.nSum <= 1) ? 3 : 2) : 0)) : a.pairMode;
#define NF_LAUNCH2(K, M, SP) hipLaunchKernelGGL((K<T, M, SP>), grid, dim3(threads), lds, stream, a)
#define NF_LAUNCH(K, M)                          \
    do {                                         \
        if (spec == 0) NF_LAUNCH2(K, M, 0);      \
        else if (spec == 1) NF_LAUNCH2(K, M, 1); \
        else NF_LAUNCH2(K, M, 2);                \
    } while (0)
    if (cols && spec == 3 && mode == 0 && sizeof(T) == 8 && threads > 512) threads = 512;   // that variant trades threads for registers
    if (cols && threads <= 256 && spec == 3) {
        if (mode == 0) NF_LAUNCH2(k_fft_cols_s, 0, 3);
        else if (mode == 1) NF_LAUNCH2(k_fft_cols_s, 1, 3);
        else NF_LAUNCH2(k_fft_cols_s, 2, 3);
    } else if (cols && threads <= 256) {
        if (mode == 0) NF_LAUNCH(k_fft_cols_s, 0);
        else if (mode == 1) NF_LAUNCH(k_fft_cols_s, 1);
        else NF_LAUNCH(k_fft_cols_s, 2);
    } else if (cols && spec == 3) {
        if (mode == 0) NF_LAUNCH2(k_fft_cols, 0, 3);
        else if (mode == 1) NF_LAUNCH2(k_fft_cols, 1, 3);
        else NF_LAUNCH2(k_fft_cols, 2, 3);
    } else if (cols) {
        if (mode == 0) NF_LAUNCH(k_fft_cols, 0);
        else if (mode == 1) NF_LAUNCH(k_fft_cols, 1);
        else NF_LAUNCH(k_fft_cols, 2);
    } else {
        if (mode == 0) NF_LAUNCH(k_fft_rows, 0);
        else if (mode == 1) NF_LAUNCH(k_fft_rows, 1);
        else NF_LAUNCH(k_fft_rows, 2);
    }
#undef NF_LAUNCH2
#undef NF_LAUNCH
    NF_CHECK_LAUNCH("k_fft_lines");
    return NF_OK;
}

static int round_up64(int v) { return (v + 63) & ~63; }

template <typename T>
int Plan2D<T>::init(int H_, int W_) {
    H = H_;
    W = W_;
    int rc = make_line_plan<T>(W, &row, &revRow);
    if (rc) return rc;
    rc = make_line_plan<T>(H, &col, &revCol);
    if (rc) return rc;
    const size_t es = sizeof(cpx<T>);
    // ---- rows: bundle rows up to ~32 KB of LDS (several workgroups per CU hide each other's latencies), at least one ----
    const size_t rowBytes = (size_t)row.mPad * es;
    if (rowBytes > (size_t)kMaxLds) {
        nf_set_error("row length too large for the single-pass LDS FFT");
        return NF_ERR_SIZE;
    }
    const size_t rowBudget = (size_t)env_int("NF_FFT_ROW_LDS_KB", 32) * 1024;
    rowsPerWG = (int)std::max<size_t>(1, std::min<size_t>(16, rowBudget / rowBytes));
    if (env_int("NF_FFT_ROWS_PER_WG", 0) > 0) rowsPerWG = env_int("NF_FFT_ROWS_PER_WG", 0);
    rowsPerWG = std::min<int>(rowsPerWG, (int)((size_t)kMaxLds / rowBytes));
    tbRow = std::max(64, std::min(256, round_up64(row.m / 8)));
    if (rowBytes > 40 * 1024 && row.m / 8 >= 512) tbRow = 512;   // at most two workgroups fit a CU: make them wider
    if (env_int("NF_FFT_ROW_TB", 0) > 0) tbRow = env_int("NF_FFT_ROW_TB", 0);
    int tl = std::max(1, std::min(rowsPerWG, std::max(1, 256 / tbRow)));
    threadsRow = tbRow * tl;
    // ---- columns: a power-of-two number of adjacent columns.  Measured on MI355X (tools/fftsweep.sh): small tiles win -
    // more resident workgroups hide the HBM and twiddle latencies, and the XCD-aware tile order lets the L2 merge the
    // short runs of neighbouring tiles - so aim at ~32 KB of LDS; long columns take what gives at least 32-byte runs ----
    const size_t colBytes = (size_t)col.mPad * es;
    if (colBytes > (size_t)kMaxLds) {
        nf_set_error("column length too large for the single-pass LDS FFT");
        return NF_ERR_SIZE;
    }
    const size_t colBudget = (size_t)env_int("NF_FFT_COL_LDS_KB", 32) * 1024;
    int c = 1;
    while (c < 32 && (size_t)(2 * c) * colBytes <= colBudget) c *= 2;
    while ((size_t)c * es < 32 && (size_t)(2 * c) * colBytes <= (size_t)kMaxLds) c *= 2;
    if (env_int("NF_FFT_COLS_PER_WG", 0) > 0) c = env_int("NF_FFT_COLS_PER_WG", 0);
    while (c > 1 && (size_t)c * colBytes > (size_t)kMaxLds) c /= 2;
    colsPerWG = c;
    threadsCol = 256;
    if ((size_t)c * colBytes > 80 * 1024) threadsCol = 1024;      // one workgroup per CU: fill it
    if (env_int("NF_FFT_COL_THREADS", 0) > 0) threadsCol = env_int("NF_FFT_COL_THREADS", 0);
    threadsCol = std::max(threadsCol, colsPerWG);
    useReal = env_int("NF_FFT_REAL", 1);
    packNyq = env_int("NF_FFT_PACK_NYQ", 1);
    tw4096 = nullptr;
    if (H == 4096 && (W % 2) == 0 && (double)H * W < 2147483648.0) {
        // twiddles of the two-pass column transform (k_cols2p_*): exp(-2 pi i k / 4096), rounded once from long double
        std::vector<cpx<T>> tw(4096);
        for (int k = 0; k < 4096; ++k) {
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / 4096.0L;
            tw[k].x = (T)cosl(ang);
            tw[k].y = (T)sinl(ang);
        }
        rc = upload<T>(tw, &tw4096);
        if (rc) return rc;
    }
    return NF_OK;
}

// Columns 0 and W/2 of a real map's half spectrum can travel through the column pass as one complex column (see k_fft_cols):
// W/2 columns are left, a whole number of tiles for the power-of-two widths whose long columns allow one workgroup per CU -
// with W/2 + 1 columns the one extra tile costs a whole extra round of the chip.  Plain-radix column plans only.
template <typename T>
bool Plan2D<T>::packOk() const {
    return packNyq && (W % 2) == 0 && W >= 4 && !col.bluestein;
}

template <typename T>
void Plan2D<T>::destroy() {
    free_line_plan<T>(&row);
    free_line_plan<T>(&col);
    if (tw4096) (void)hipFree((void*)tw4096);
    tw4096 = nullptr;
}

template <typename T>
static LineArgs<T> row_args(const Plan2D<T>& P, const void* in, void* out, int inType, int outType) {
    LineArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.plan = P.row;
    a.in = in;
    a.out = out;
    a.inType = inType;
    a.outType = outType;
    a.inElemStride = 1;
    a.inLineStride = P.W;
    a.inBatchStride = (long long)P.H * P.W;
    a.outElemStride = 1;
    a.outLineStride = P.W;
    a.outBatchStride = (long long)P.H * P.W;
    a.conjInSign = a.conjOutSign = a.scale = 1.0;
    a.realH = P.H;
    a.realW = P.W;
    a.nLines = P.H;
    a.linesPerWG = P.rowsPerWG;
    a.tb = P.tbRow;
    return a;
}

template <typename T>
static LineArgs<T> col_args(const Plan2D<T>& P, const void* in, void* out) {
    LineArgs<T> c;
    memset(&c, 0, sizeof(c));
    c.plan = P.col;
    c.in = in;
    c.out = out;
    c.inType = 0;
    c.outType = 0;
    c.inElemStride = P.W;
    c.inLineStride = 1;
    c.inBatchStride = (long long)P.H * P.W;
    c.outElemStride = P.W;
    c.outLineStride = 1;
    c.outBatchStride = (long long)P.H * P.W;
    c.conjInSign = c.conjOutSign = c.scale = 1.0;
    c.realH = P.H;
    c.realW = P.W;
    c.nLines = P.W;
    c.linesPerWG = P.colsPerWG;
    c.nTiles = nf_div_up(P.W, P.colsPerWG);
    return c;
}

// forward: in (real f32/f64 or complex<T>) -> out complex<T> [batch][H][W]; optional apodisation of real input;
// result divided by `scale` if scaleDiv (enmap.fft: sqrt(H*W)).
template <typename T>
int Plan2D<T>::forward(const void* in, int inType, void* out, int batch, const double* apodY, const double* apodX,
                       int scaleDiv, double scale, hipStream_t stream, int half, int rowsOnly) {
    LineArgs<T> a = row_args<T>(*this, in, out, inType, 0);
    a.apodLine = apodY;
    a.apodElem = apodX;
    LineArgs<T> c = col_args<T>(*this, out, out);
    if (scaleDiv) c.scale = 1.0 / scale;
    if (half && inType == 0) {
        nf_set_error("half-stored spectra are spectra of real maps: the input must be real");
        return NF_ERR_ARG;
    }
    if (inType != 0 && (useReal || half)) {
        // real map: two rows per complex line, half of the columns; the other half by conjugate symmetry, or (half) left
        // unwritten for consumers that know F[-y][-c] = conj F[y][c]
        a.pairMode = 1;
        a.nLines = (H + 1) / 2;
        c.mirror = half ? 0 : 1;
        c.nLines = W / 2 + 1;
        c.nTiles = nf_div_up(c.nLines, colsPerWG);
    }
    if (a.pairMode == 1 && packOk()) {
        c.packNyq = 1;
        c.nLines = W / 2;
        c.nTiles = nf_div_up(c.nLines, colsPerWG);
    }
    int rc = launch_lines<T>(a, false, batch, threadsRow, stream);
    if (rc || rowsOnly) return rc;
    return launch_lines<T>(c, true, batch, threadsCol, stream);
}

// inverse (unnormalised backward sum, then optional division by `scale`): complex<T> in -> work (complex, same
// size, may alias `in` if the caller allows clobbering) -> out (complex<T>, or real part as f64 / f32).
template <typename T>
int Plan2D<T>::inverse(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                       hipStream_t stream, const float* filt, int nSum, long long filtBatchStride, const double* winY,
                       const double* winX) {
    LineArgs<T> c = col_args<T>(*this, in, work);
    c.conjInSign = -1.0;  // conj(FFT(conj x)): the conj between the two passes cancels, so only the ends matter
    c.mulF32 = filt;
    c.mulBatchStride = filtBatchStride;
    c.nSum = nSum < 1 ? 1 : nSum;
    c.sumStride = (long long)H * W;
    c.winLine = winX;
    c.winElem = winY;
    c.fused = (filt != nullptr || winY != nullptr) ? 1 : 0;
    LineArgs<T> a = row_args<T>(*this, work, out, 0, outType);
    a.conjOutSign = -1.0;
    if (scaleDiv) a.scale = 1.0 / scale;
    if (outType != 0 && useReal) {
        // only the real part is wanted: transform the symmetrised spectrum - half of the columns, two rows per line.
        // The half plane left in `work` is conj(column-inverse), which is Hermitian along x like the column-inverse
        // itself; packed as Z = V1 + i V2 its forward row transform is x1 + i x2 with no conjugation at either end.
        c.herm = 1;
        c.fused = 1;
        c.nLines = W / 2 + 1;
        c.nTiles = nf_div_up(c.nLines, colsPerWG);
        a.pairMode = 2;
        a.nLines = (H + 1) / 2;
        a.conjOutSign = 1.0;
    }
    int rc = launch_lines<T>(c, true, batch, threadsCol, stream);
    if (rc) return rc;
    return launch_lines<T>(a, false, batch, threadsRow, stream);
}

template <typename T>
int Plan2D<T>::inverse_half(const void* in, void* work, void* out, int outType, int batch, int scaleDiv, double scale,
                            hipStream_t stream, const double* filtSym, int nSum, long long filtBatchStride, const double* winY,
                            const double* winX, const double* epiScale, const unsigned char* epiMask, const void* epiZeroRef,
                            const double* planeScale, int rowLo, int rowHi, long long inBatchStride, long long epiScaleStride) {
    if (outType == 0) {
        nf_set_error("a half-stored spectrum has a real inverse transform: ask for real output");
        return NF_ERR_ARG;
    }
    LineArgs<T> c = col_args<T>(*this, in, work);
    if (inBatchStride >= 0) c.inBatchStride = inBatchStride;      // (0: every batch entry transforms the same planes, each with its filter)
    c.conjInSign = -1.0;
    c.mulSym = filtSym;
    c.mulBatchStride = filtBatchStride;
    c.nSum = nSum < 1 ? 1 : nSum;
    c.sumStride = (long long)H * W;
    c.winLine = winX;
    c.winElem = winY;
    c.herm = 1;
    c.halfIn = 1;
    c.fused = 1;
    if (planeScale != nullptr) {
        const int np_ = nSum > 1 ? nSum : batch;
        if (np_ > NF_PLANE_SCALES) {
            nf_set_error("inverse_half: at most 16 plane factors");
            return NF_ERR_ARG;
        }
        c.usePlaneScale = 1;
        for (int i = 0; i < np_; ++i) c.planeScale[i] = planeScale[i];
    }
    c.nLines = W / 2 + 1;
    c.nTiles = nf_div_up(c.nLines, colsPerWG);
    LineArgs<T> a = row_args<T>(*this, work, out, 0, outType);
    if (scaleDiv) a.scale = 1.0 / scale;
    a.pairMode = 2;
    a.nLines = (H + 1) / 2;
    a.conjOutSign = 1.0;
    a.epiScale = epiScale;
    a.epiScaleStride = epiScaleStride;
    a.epiMask = epiMask;
    a.epiZeroRef = epiZeroRef;
    if (packOk()) {
        c.packNyq = 1;
        c.nLines = W / 2;
        c.nTiles = nf_div_up(c.nLines, colsPerWG);
    }
    if (rowHi > rowLo) {
        // only the output rows rowLo .. rowHi - 1 are wanted (whole row pairs): the column pass stores just those rows of the
        // intermediate plane, the row pass runs on just those lines
        const int lo = (rowLo < 0 ? 0 : rowLo) & ~1, hi = rowHi >= H ? H : ((rowHi + 1) & ~1);
        if (lo < hi && (lo > 0 || hi < H)) {
            c.storeLo = lo;
            c.storeHi = hi;
            a.lineBase = lo / 2;
            a.nLines = (hi + 1) / 2;
        }
    }
    int rc;
    static const int twoPass = env_int("NF_FFT_COLS_2PASS", 1);
    if (twoPass && tw4096 != nullptr && c.nSum <= 1 && c.storeHi == 0) {
        // 4096-long columns: two passes of 64-point transforms (k_cols2p_first / _second), rows of `work` left in swapped order
        c.nTiles = nf_div_up(c.nLines, NF_C2P_CT);
        const double npixTotal = ((double)c.nLines + (c.packNyq ? 1.0 : 0.0)) * c.plan.n * batch;
        ProfScope ps(sizeof(T) == 8 ? PT_FFT_COLS_F64 : PT_FFT_COLS_F32, stream,
                     npixTotal * (2.0 * sizeof(cpx<T>) + (c.mulSym ? 8.0 : 0.0)));
        const dim3 grid((unsigned)c.nTiles * 64u, (unsigned)batch);
        hipLaunchKernelGGL((k_cols2p_first<T>), grid, dim3(NF_C2P_CT * 8), 0, stream, c, tw4096);
        NF_CHECK_LAUNCH("k_cols2p_first");
        hipLaunchKernelGGL((k_cols2p_second<T>), grid, dim3(NF_C2P_CT * 8), 0, stream, c, tw4096);
        NF_CHECK_LAUNCH("k_cols2p_second");
        a.rowSwap64 = 1;
        rc = NF_OK;
    } else {
        rc = launch_lines<T>(c, true, batch, threadsCol, stream);
    }
    if (rc) return rc;
    return launch_lines<T>(a, false, batch, threadsRow, stream);
}

template <typename T>
bool Plan2D<T>::template_filter_ok(int batch) const {
    static const int on = env_int("NF_TEMPLATE_FUSED", 1);
    if (!on || sizeof(T) != 4 || col.bluestein || row.bluestein || batch < 1 || batch > NF_TF_MAXBATCH) return false;
    LineArgs<T> c;
    memset(&c, 0, sizeof(c));
    c.linesPerWG = colsPerWG;
    c.nLines = packOk() ? W / 2 : W / 2 + 1;
    c.nTiles = nf_div_up(c.nLines, colsPerWG);
    fit_col_tiles<T>(c, batch);
    const long long share = ((long long)col.n * c.linesPerWG + threadsCol - 1) / threadsCol;
    return share <= NF_TF_MAXE && threadsCol <= NF_FFT_MAXTHREADS && (threadsCol % c.linesPerWG) == 0;
}

template <typename T>
int Plan2D<T>::template_filter(const void* rowsOut, int batch, const TemplateFuse& tf, int covType, const double* planeScale,
                               const double* winY, const double* winX, void* work, void* out, int outType, int rowLo, int rowHi,
                               int scaleDiv, double scale, hipStream_t stream) {
    if constexpr (sizeof(T) != 4) {
        nf_set_error("template_filter: float32 templates only");
        return NF_ERR_ARG;
    } else {
        if (!template_filter_ok(batch) || outType == 0 || (planeScale != nullptr && batch > 8)) {
            nf_set_error("template_filter: not available for this plan");
            return NF_ERR_ARG;
        }
        // the forward column pass as forward(half = 1) sets it up
        LineArgs<float> cf = col_args<float>(*this, rowsOut, const_cast<void*>(rowsOut));
        if (scaleDiv) cf.scale = 1.0 / scale;
        cf.nLines = W / 2 + 1;
        if (packOk()) {
            cf.packNyq = 1;
            cf.nLines = W / 2;
        }
        cf.nTiles = nf_div_up(cf.nLines, colsPerWG);
        // the inverse column pass and the row pass as inverse_half(batch entries with one plane each) sets them up
        LineArgs<float> ci = col_args<float>(*this, rowsOut, work);
        ci.conjInSign = -1.0;
        ci.mulSym = tf.sym;
        ci.mulBatchStride = (long long)H * W;
        ci.nSum = 1;
        ci.sumStride = (long long)H * W;
        ci.winLine = winX;
        ci.winElem = winY;
        ci.herm = 1;
        ci.halfIn = 1;
        ci.fused = 1;
        if (planeScale != nullptr) {
            ci.usePlaneScale = 1;
            for (int i = 0; i < batch; ++i) ci.planeScale[i] = planeScale[i];
        }
        ci.nLines = cf.nLines;
        ci.nTiles = cf.nTiles;
        ci.packNyq = cf.packNyq;
        LineArgs<float> a = row_args<float>(*this, work, out, 0, outType);
        a.pairMode = 2;
        a.nLines = (H + 1) / 2;
        a.conjOutSign = 1.0;
        if (rowHi > rowLo) {
            const int lo = (rowLo < 0 ? 0 : rowLo) & ~1, hi = rowHi >= H ? H : ((rowHi + 1) & ~1);
            if (lo < hi && (lo > 0 || hi < H)) {
                ci.storeLo = lo;
                ci.storeHi = hi;
                a.lineBase = lo / 2;
                a.nLines = (hi + 1) / 2;
            }
        }
        fit_col_tiles<float>(cf, batch);
        ci.linesPerWG = cf.linesPerWG;
        ci.nTiles = cf.nTiles;
        const size_t lds = (size_t)cf.plan.mPad * cf.linesPerWG * sizeof(cpx<float>);
        if (lds > (size_t)kMaxLds || (double)(W / 2 + 1) * cf.plan.n >= 2147483648.0) {
            nf_set_error("template_filter: column tile does not fit");
            return NF_ERR_SIZE;
        }
        static const int pairOn = env_int("NF_FFT_COL_PAIR", 1);
        const bool pairBase = pairOn && lds > 80 * 1024 && (cf.linesPerWG & 1) == 0 && (W & 1) == 0 && (((long long)H * W) & 1) == 0;
        cf.colPair = pairBase ? 1 : 0;
        ci.colPair = pairBase ? 1 : 0;
        int mode = 0;
        for (int s = 0; s < cf.plan.nst; ++s)
            if (cf.plan.st[s].R > 8) mode = 1;
        static bool attrSet = false;
        if (!attrSet) {
#define NF_ATTR(K) NF_HIP(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds))
            NF_ATTR((k_template_cols<double, 0>));
            NF_ATTR((k_template_cols<double, 1>));
            NF_ATTR((k_template_cols<float, 0>));
            NF_ATTR((k_template_cols<float, 1>));
#undef NF_ATTR
            attrSet = true;
        }
        int nCubes = 0;
        for (int b = 0; b < batch; ++b) nCubes += tf.filt[b] != nullptr;
        const double covB = covType == NF_F32 ? 4.0 : 8.0;
        static const int threePass = env_int("NF_TEMPLATE_2PASS", 1);
        if (threePass && tw4096 != nullptr && cf.packNyq) {
            // 4096-long columns: the path as three streaming passes (k_template2p_*), see there
            cf.nTiles = nf_div_up(cf.nLines, NF_C2P_CT);
            ci.nTiles = cf.nTiles;
            ProfScope ps(PT_FFT_COLS_F32, stream,
                         (double)H * W * (batch * (4.0 + covB + 4.0) + nCubes * 4.0) + (double)(ci.storeHi > 0 ? ci.storeHi - ci.storeLo : H) * (W / 2 + 1) * 8.0 * batch);
            const dim3 grid((unsigned)cf.nTiles * 64u, (unsigned)batch), block(NF_C2P_CT * 8);
            const long long wbs = (long long)H * W;
            hipLaunchKernelGGL(k_template2p_first, grid, block, 0, stream, cf, (cpx<float>*)work, wbs, (const cpx<float>*)tw4096);
            NF_CHECK_LAUNCH("k_template2p_first");
            if (covType == NF_F64)
                hipLaunchKernelGGL((k_template2p_middle<double>), grid, block, 0, stream, cf, ci, tf, (cpx<float>*)work, wbs, (const cpx<float>*)tw4096);
            else
                hipLaunchKernelGGL((k_template2p_middle<float>), grid, block, 0, stream, cf, ci, tf, (cpx<float>*)work, wbs, (const cpx<float>*)tw4096);
            NF_CHECK_LAUNCH("k_template2p_middle");
            hipLaunchKernelGGL(k_template2p_last, grid, block, 0, stream, ci, (cpx<float>*)work, wbs, (const cpx<float>*)tw4096);
            NF_CHECK_LAUNCH("k_template2p_last");
        } else {
            // algorithmic bytes: the row spectra and the noise power in; the symmetrised filter, the wanted cubes and the stored rows out
            ProfScope ps(PT_FFT_COLS_F32, stream,
                         (double)H * W * (batch * (4.0 + covB + 4.0) + nCubes * 4.0) + (double)(ci.storeHi > 0 ? ci.storeHi - ci.storeLo : H) * (W / 2 + 1) * 8.0 * batch);
            dim3 grid(((cf.nTiles + 7) / 8) * 8, batch);
            if (covType == NF_F64) {
                if (mode == 0)
                    hipLaunchKernelGGL((k_template_cols<double, 0>), grid, dim3(threadsCol), lds, stream, cf, ci, tf);
                else
                    hipLaunchKernelGGL((k_template_cols<double, 1>), grid, dim3(threadsCol), lds, stream, cf, ci, tf);
            } else {
                if (mode == 0)
                    hipLaunchKernelGGL((k_template_cols<float, 0>), grid, dim3(threadsCol), lds, stream, cf, ci, tf);
                else
                    hipLaunchKernelGGL((k_template_cols<float, 1>), grid, dim3(threadsCol), lds, stream, cf, ci, tf);
            }
            NF_CHECK_LAUNCH("k_template_cols");
        }
        return launch_lines<float>(a, false, batch, threadsRow, stream);
    }
}

// y = IFFT(FFT(x) * w) along one axis of a real map (k_line_filter): axis 0 = along the rows (x), 1 = along the columns (y).
// wperm: device table of n doubles, wperm[rev[k]] = w[k] / n.  in == out is allowed for axis 1 only (a tile is read whole
// before it is written; a rows workgroup would overwrite rows other workgroups still read... no: rows are disjoint too).
template <typename T>
int Plan2D<T>::axis_filter(int axis, const void* in, int inType, void* out, int outType, int batch, const double* wperm,
                           const void* zeroRef, hipStream_t stream, const int* zeroRows) {
    const LinePlan<T>& lp = axis == 0 ? row : col;
    if (lp.bluestein) {
        nf_set_error("axis_filter: this length needs Bluestein's algorithm - use the 2-D transforms");
        return NF_ERR_SIZE;
    }
    LineArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.plan = lp;
    a.in = in;
    a.out = out;
    a.inType = inType;
    a.outType = outType;
    a.realH = H;
    a.realW = W;
    a.winElem = wperm;
    a.epiZeroRef = zeroRef;
    a.epiZeroRows = (axis == 0 && zeroRef != nullptr) ? zeroRows : nullptr;          // [batch][H]
    a.conjInSign = a.conjOutSign = a.scale = 1.0;
    int mode = 0;
    for (int s = 0; s < lp.nst; ++s)
        if (lp.st[s].R > 8) mode = 1;
    static bool attrSet[2] = {false, false};
    const int which = sizeof(T) == 8 ? 1 : 0;
    if (!attrSet[which]) {
#define NF_ATTR(K) NF_HIP(hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds))
        NF_ATTR((k_line_filter<T, false, 0>));
        NF_ATTR((k_line_filter<T, false, 1>));
        NF_ATTR((k_line_filter<T, true, 0>));
        NF_ATTR((k_line_filter<T, true, 1>));
#undef NF_ATTR
        attrSet[which] = true;
    }
    const double bytes = (double)H * W * batch * ((inType == 1 ? 8.0 : 4.0) + (outType == 1 ? 8.0 : 4.0) * ((zeroRef && !a.epiZeroRows) ? 2.0 : 1.0));
    if (axis == 0) {
        a.nLines = (H + 1) / 2;
        a.linesPerWG = rowsPerWG;
        a.tb = tbRow;
        const size_t lds = (size_t)lp.mPad * a.linesPerWG * sizeof(cpx<T>);
        ProfScope ps(PT_AXIS_FILTER_ROWS, stream, bytes);
        dim3 grid(nf_div_up(a.nLines, a.linesPerWG), batch);
        static const int fusedRows = env_int("NF_LF_FUSED", 1);
        if (mode == 0 && lp.nst >= 2 && fusedRows) {
            const int rcF = launch_line_filter_rows_fused<T>(a, batch, threadsRow, lds, stream);
            if (rcF) return rcF;
        } else if (mode == 0)
            hipLaunchKernelGGL((k_line_filter<T, false, 0>), grid, dim3(threadsRow), lds, stream, a);
        else
            hipLaunchKernelGGL((k_line_filter<T, false, 1>), grid, dim3(threadsRow), lds, stream, a);
    } else {
        a.nLines = (W + 1) / 2;
        a.linesPerWG = colsPerWG;
        a.nTiles = nf_div_up(a.nLines, colsPerWG);
        fit_col_tiles<T>(a, batch);
        const size_t lds = (size_t)lp.mPad * a.linesPerWG * sizeof(cpx<T>);
        ProfScope ps(PT_AXIS_FILTER_COLS, stream, bytes);
        dim3 grid(((a.nTiles + 7) / 8) * 8, batch);
        static const int lfThreads = env_int("NF_LF_COL_THREADS", 0);
        const int thr = lfThreads > 0 ? std::max(lfThreads, a.linesPerWG) : threadsCol;
        static const int fusedCols = env_int("NF_LF_FUSED_COLS", 1);
        if (mode == 0 && lp.nst >= 2 && fusedCols) {
            const int rcF = launch_line_filter_cols_fused<T>(a, batch, thr, lds, stream);
            if (rcF) return rcF;
        } else if (mode == 0)
            hipLaunchKernelGGL((k_line_filter<T, true, 0>), grid, dim3(thr), lds, stream, a);
        else
            hipLaunchKernelGGL((k_line_filter<T, true, 1>), grid, dim3(thr), lds, stream, a);
    }
    NF_CHECK_LAUNCH("k_line_filter");
    return NF_OK;
}

template struct Plan2D<float>;
template struct Plan2D<double>;

}  // namespace nf

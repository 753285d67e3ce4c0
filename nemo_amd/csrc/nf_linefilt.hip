// y = IFFT(FFT(x) * w) along one axis of a real map (enmap.apply_window's transform pair along x or y, nemo/filters.py:100,644)
// with the stages fused at both ends and in the middle: k_line_filter_rows_fused / k_line_filter_cols_fused, launched by
// Plan2D::axis_filter (nf_fft.hip) in place of k_line_filter (NF_LF_FUSED=0 / NF_LF_FUSED_COLS=0 restore that kernel).
//
// k_line_filter copies the two real lines of a complex line into LDS, runs the forward stages LDS -> LDS, multiplies in a pass
// of its own, runs the inverse stages and leaves through a store pass: with S stages the line crosses the LDS 4 S + 4 times
// between 2 S + 3 barriers.  Here
//   * the FIRST forward stage (decimation in frequency, stride n / R: natural order - along a row consecutive lanes take
//     consecutive pixels, along columns a lane takes its two adjacent columns as one 16-byte access) reads its operands
//     straight from the map in global memory,
//   * the LAST forward stage, the product with the transfer function and the FIRST inverse stage are one step: both stages
//     have stride 1, i.e. work on the same R neighbouring (digit-reversed) places,
//   * the LAST inverse stage stores the two real lines straight from its registers (natural order again),
// so the line crosses the LDS 4 S - 4 times between 2 S - 2 barriers (4096 = 8^4: 12 crossings instead of 20).
// Per element the arithmetic is k_line_filter's, statement for statement (same butterflies, twiddles and roundings; the
// product with w is kept out of the butterflies' fused multiply-adds, as its LDS round trip kept it there): the same bits
// (tests/test_gpu_kernels.py::test_fused_line_filter_equals_staged_kernel_bit_for_bit).  Measured on MI355X
// (tools/linefilt_ab.sh): rows 4096^2 float64 127 -> 104 us, 8 x 840 x 1440 92 -> 82; columns 131 -> 113 and 67 -> 56.
// Plain radices (2, 3, 4, 5, 7, 8) and at least two stages; anything else stays with k_line_filter.
#include "nf_common.h"
#include "nf_fft.h"

#pragma clang fp contract(fast)

#include "nf_fft_dev.h"

namespace {

template <typename T>
__device__ __forceinline__ T lff_load(const void* p, bool f64, size_t off) {
    return f64 ? (T)((const double*)p)[off] : (T)((const float*)p)[off];
}

// DIF stage `st` (the plan's last: L = n / R) with operands from the two real rows; results twiddled into LDS
template <typename T, int R>
__device__ __forceinline__ void lff_first(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R, nb = (unsigned)p.m / R;
    const cpx<T>* __restrict__ twp = p.stw + st.twOff;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    const size_t W = (size_t)a.realW;
    const bool f64 = a.inType == 1;
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
        for (int line = G.tl; line < G.nl; line += G.TL) {
            const size_t rowA = 2u * (size_t)(G.line0 + line);
            const bool haveB = rowA + 1 < (size_t)a.realH;
            const size_t oa = rowA * W + base, ob = (haveB ? rowA + 1 : rowA) * W + base;
            cpx<T>* __restrict__ lb = buf + (unsigned)line * (unsigned)p.mPad;
            cpx<T> x[R];
            // (one uniform branch around the whole batch of loads: a select per load would put a wait behind each)
            if (f64) {
                double va[R], vb[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    va[k] = ((const double*)G.inP)[oa + (size_t)k * L];
                    vb[k] = ((const double*)G.inP)[ob + (size_t)k * L];
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[k].x = (T)va[k];
                    x[k].y = haveB ? (T)vb[k] : (T)0;
                }
            } else {
                float va[R], vb[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    va[k] = ((const float*)G.inP)[oa + (size_t)k * L];
                    vb[k] = ((const float*)G.inP)[ob + (size_t)k * L];
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[k].x = (T)va[k];
                    x[k].y = haveB ? (T)vb[k] : (T)0;
                }
            }
            Dft<T, R>::run(x, wR);
            if (L > 1) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) lb[(unsigned)swz((int)(base + (unsigned)k * L))] = x[k];
        }
    }
    __syncthreads();
}

// one stage LDS -> LDS (fft_stage of nf_fft.hip without its global-memory and special-case paths)
template <typename T, int R, bool COLS>
__device__ __forceinline__ void lff_stage(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const bool dif,
                                          const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R, nb = (unsigned)p.m / R;
    const unsigned lineMul = COLS ? 1u : (unsigned)p.mPad, elemMul = COLS ? (unsigned)G.TL : 1u;   // LDS: line * lineMul + swz(e) * elemMul
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
        unsigned idx[R];
#pragma unroll
        for (int k = 0; k < R; ++k) idx[k] = (unsigned)swz((int)(base + (unsigned)k * L)) * elemMul;
        for (int line = G.tl; line < G.nl; line += (COLS ? G.nl : G.TL)) {
            cpx<T>* __restrict__ lb = buf + (unsigned)line * lineMul;
            cpx<T> x[R];
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lb[idx[k]];
            if (L > 1 && !dif) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
            Dft<T, R>::run(x, wR);
            if (L > 1 && dif) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) lb[idx[k]] = x[k];
        }
    }
    __syncthreads();
}

// the rounded product of the spectrum with the transfer function, conjugated (the way back is a forward transform of the
// conjugate): kept apart from the butterflies' contractions
template <typename T>
__device__ __forceinline__ cpx<T> lff_times(cpx<T> v, T w) {
#pragma clang fp contract(off)
    cpx<T> r;
    r.x = v.x * w;
    r.y = -(v.y * w);
    return r;
}

// stage 0 forward (DIF, stride 1), times w, stage 0 back (DIT, stride 1): the same R places
template <typename T, int R, bool COLS>
__device__ __forceinline__ void lff_middle(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned nb = (unsigned)p.m / R;
    const unsigned lineMul = COLS ? 1u : (unsigned)p.mPad, elemMul = COLS ? (unsigned)G.TL : 1u;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    for (unsigned b = (unsigned)G.tb; b < nb; b += (unsigned)G.TB) {
        const unsigned base = b * R;
        unsigned idx[R];
        T wv[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            idx[k] = (unsigned)swz((int)(base + (unsigned)k)) * elemMul;
            wv[k] = (T)a.winElem[base + (unsigned)k];
        }
        for (int line = G.tl; line < G.nl; line += (COLS ? G.nl : G.TL)) {
            cpx<T>* __restrict__ lb = buf + (unsigned)line * lineMul;
            cpx<T> x[R];
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lb[idx[k]];
            Dft<T, R>::run(x, wR);
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lff_times<T>(x[k], wv[k]);
            Dft<T, R>::run(x, wR);
#pragma unroll
            for (int k = 0; k < R; ++k) lb[idx[k]] = x[k];
        }
    }
    __syncthreads();
}

// DIT stage `st` (the plan's last) with the two real rows stored from its registers: real part and minus the imaginary part;
// where the zero reference (a map of the output type) is zero the output is zero (nemo/filters.py:99-101)
template <typename T, int R>
__device__ __forceinline__ void lff_last(const cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const Geo& G,
                                         const void* refP, const int* __restrict__ zeroRows) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R, nb = (unsigned)p.m / R;
    const cpx<T>* __restrict__ twp = p.stw + st.twOff;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    const size_t W = (size_t)a.realW;
    const bool o64 = a.outType == 1;
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
        for (int line = G.tl; line < G.nl; line += G.TL) {
            const size_t rowA = 2u * (size_t)(G.line0 + line);
            const bool haveB = rowA + 1 < (size_t)a.realH;
            const cpx<T>* __restrict__ lb = buf + (unsigned)line * (unsigned)p.mPad;
            // (a row the caller knows to hold no zero is not looked up in the zero reference)
            const bool refA = refP != nullptr && (zeroRows == nullptr || zeroRows[rowA] != 0);
            const bool refB = haveB && refP != nullptr && (zeroRows == nullptr || zeroRows[rowA + 1] != 0);
            const size_t oa = rowA * W + base, ob = (rowA + 1) * W + base;
            cpx<T> x[R];
#pragma unroll
            for (int k = 0; k < R; ++k) x[k] = lb[(unsigned)swz((int)(base + (unsigned)k * L))];
            if (L > 1) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
            Dft<T, R>::run(x, wR);
            if (o64) {
                double za[R], zb[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    za[k] = refA ? ((const double*)refP)[oa + (size_t)k * L] : 1.0;
                    zb[k] = refB ? ((const double*)refP)[ob + (size_t)k * L] : 1.0;
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    ((double*)G.outP)[oa + (size_t)k * L] = za[k] == 0.0 ? 0.0 : (double)x[k].x;
                    if (haveB) ((double*)G.outP)[ob + (size_t)k * L] = zb[k] == 0.0 ? 0.0 : (double)-x[k].y;
                }
            } else {
                float za[R], zb[R];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    za[k] = refA ? ((const float*)refP)[oa + (size_t)k * L] : 1.0f;
                    zb[k] = refB ? ((const float*)refP)[ob + (size_t)k * L] : 1.0f;
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    ((float*)G.outP)[oa + (size_t)k * L] = za[k] == 0.0f ? 0.0f : (float)x[k].x;
                    if (haveB) ((float*)G.outP)[ob + (size_t)k * L] = zb[k] == 0.0f ? 0.0f : (float)-x[k].y;
                }
            }
        }
    }
}

#define NF_LFF_SWITCH(CALL)          \
    switch (st.R) {                  \
        case 2: CALL(2); break;      \
        case 3: CALL(3); break;      \
        case 4: CALL(4); break;      \
        case 5: CALL(5); break;      \
        case 7: CALL(7); break;      \
        case 8: CALL(8); break;      \
        default: break;              \
    }

template <typename T>
__global__ void __launch_bounds__(512) k_line_filter_rows_fused(nf::LineArgs<T> a) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    cpx<T>* buf = (cpx<T>*)nf_smem;
    const nf::LinePlan<T>& p = a.plan;
    Geo G;
    G.line0 = blockIdx.x * a.linesPerWG;
    G.nl = min(a.linesPerWG, a.nLines - G.line0);
    G.TB = a.tb;
    G.TL = blockDim.x / a.tb;
    G.tl = threadIdx.x / a.tb;
    G.tb = threadIdx.x - G.tl * a.tb;
    const size_t plane = (size_t)a.realH * (size_t)a.realW;
    const size_t inEs = a.inType == 1 ? 8 : 4, outEs = a.outType == 1 ? 8 : 4;
    G.inP = (const char*)a.in + (size_t)blockIdx.y * plane * inEs;
    G.outP = (char*)a.out + (size_t)blockIdx.y * plane * outEs;
    // (every map of a batch has its own zero reference and its own row flags, laid out like the maps)
    const void* refP = a.epiZeroRef != nullptr ? (const void*)((const char*)a.epiZeroRef + (size_t)blockIdx.y * plane * outEs) : nullptr;
    const int* __restrict__ zeroRows = a.epiZeroRows != nullptr ? a.epiZeroRows + (size_t)blockIdx.y * (size_t)a.realH : nullptr;
    const int S = p.nst;
    {
        const nf::StageInfo st = p.st[S - 1];
#define NF_LFF_FIRST(R) lff_first<T, R>(buf, a, st, G)
        NF_LFF_SWITCH(NF_LFF_FIRST)
#undef NF_LFF_FIRST
    }
    for (int s = S - 2; s >= 1; --s) {
        const nf::StageInfo st = p.st[s];
#define NF_LFF_DIF(R) lff_stage<T, R, false>(buf, a, st, true, G)
        NF_LFF_SWITCH(NF_LFF_DIF)
#undef NF_LFF_DIF
    }
    {
        const nf::StageInfo st = p.st[0];
#define NF_LFF_MID(R) lff_middle<T, R, false>(buf, a, st, G)
        NF_LFF_SWITCH(NF_LFF_MID)
#undef NF_LFF_MID
    }
    for (int s = 1; s <= S - 2; ++s) {
        const nf::StageInfo st = p.st[s];
#define NF_LFF_DIT(R) lff_stage<T, R, false>(buf, a, st, false, G)
        NF_LFF_SWITCH(NF_LFF_DIT)
#undef NF_LFF_DIT
    }
    {
        const nf::StageInfo st = p.st[S - 1];
#define NF_LFF_LAST(R) lff_last<T, R>(buf, a, st, G, refP, zeroRows)
        NF_LFF_SWITCH(NF_LFF_LAST)
#undef NF_LFF_LAST
    }
}

// ---- columns: lane tl of a tile = columns 2 c, 2 c + 1 (c = line0 + tl), LDS layout [swz(e)][tl] ------------------------------
// first DIF stage from global memory: the lane's two columns as one 2-vector where the pitch and the pointer allow it
template <typename T, int R>
__device__ __forceinline__ void lffc_first(cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const Geo& G) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R, nb = (unsigned)p.m / R;
    const cpx<T>* __restrict__ twp = p.stw + st.twOff;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    const size_t W = (size_t)a.realW;
    const bool f64 = a.inType == 1;
    const size_t c0 = 2u * (size_t)(G.line0 + G.tl);
    const bool haveB = c0 + 1 < W;
    const bool vec = (W & 1) == 0 && ((size_t)G.inP & (f64 ? 15 : 7)) == 0;
    if (G.nl > 0) {
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
            const size_t o0 = (size_t)base * W + c0, dO = (size_t)L * W;
            cpx<T> x[R];
            if (f64 && vec) {
                double2 v[R];
#pragma unroll
                for (int k = 0; k < R; ++k) v[k] = *(const double2*)((const double*)G.inP + o0 + (size_t)k * dO);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[k].x = (T)v[k].x;
                    x[k].y = (T)v[k].y;
                }
            } else if (vec) {
                float2 v[R];
#pragma unroll
                for (int k = 0; k < R; ++k) v[k] = *(const float2*)((const float*)G.inP + o0 + (size_t)k * dO);
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[k].x = (T)v[k].x;
                    x[k].y = (T)v[k].y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    x[k].x = lff_load<T>(G.inP, f64, o0 + (size_t)k * dO);
                    x[k].y = haveB ? lff_load<T>(G.inP, f64, o0 + (size_t)k * dO + 1) : (T)0;
                }
            }
            Dft<T, R>::run(x, wR);
            if (L > 1) {
#pragma unroll
                for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) buf[(unsigned)swz((int)(base + (unsigned)k * L)) * (unsigned)G.TL + (unsigned)G.tl] = x[k];
        }
    }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ void lffc_store(const nf::LineArgs<T>& a, void* outP, const void* refP, size_t off, T v) {
    if (a.outType == 1) {
        double w = (double)v;
        if (refP != nullptr && ((const double*)refP)[off] == 0.0) w = 0.0;
        ((double*)outP)[off] = w;
    } else {
        float w = (float)v;
        if (refP != nullptr && ((const float*)refP)[off] == 0.0f) w = 0.0f;
        ((float*)outP)[off] = w;
    }
}

// last DIT stage stores the two real columns from its registers
template <typename T, int R>
__device__ __forceinline__ void lffc_last(const cpx<T>* __restrict__ buf, const nf::LineArgs<T>& a, const nf::StageInfo st, const Geo& G,
                                          const void* refP) {
    const nf::LinePlan<T>& p = a.plan;
    const unsigned L = (unsigned)st.L, LR = L * R, nb = (unsigned)p.m / R;
    const cpx<T>* __restrict__ twp = p.stw + st.twOff;
    cpx<T> wR[R];
    if (R != 2 && R != 4 && R != 8 && R != 3) {
#pragma unroll
        for (int j = 0; j < R; ++j) wR[j] = p.stw[st.rootOff + j];
    }
    if (G.nl <= 0) return;
    const size_t W = (size_t)a.realW;
    const bool o64 = a.outType == 1;
    const size_t c0 = 2u * (size_t)(G.line0 + G.tl);
    const bool haveB = c0 + 1 < W;
    const bool vec = refP == nullptr && (W & 1) == 0 && ((size_t)G.outP & (o64 ? 15 : 7)) == 0;
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
        cpx<T> x[R];
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = buf[(unsigned)swz((int)(base + (unsigned)k * L)) * (unsigned)G.TL + (unsigned)G.tl];
        if (L > 1) {
#pragma unroll
            for (int k = 1; k < R; ++k) x[k] = cmul(x[k], w[k - 1]);
        }
        Dft<T, R>::run(x, wR);
        const size_t o0 = (size_t)base * W + c0, dO = (size_t)L * W;
        if (vec && o64) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                double2 o;
                o.x = (double)x[k].x;
                o.y = (double)-x[k].y;
                *(double2*)((double*)G.outP + o0 + (size_t)k * dO) = o;
            }
        } else if (vec) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                float2 o;
                o.x = (float)x[k].x;
                o.y = (float)-x[k].y;
                *(float2*)((float*)G.outP + o0 + (size_t)k * dO) = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                lffc_store<T>(a, G.outP, refP, o0 + (size_t)k * dO, x[k].x);
                if (haveB) lffc_store<T>(a, G.outP, refP, o0 + (size_t)k * dO + 1, -x[k].y);
            }
        }
    }
}

template <typename T, int THREADS>
__global__ void __launch_bounds__(THREADS) k_line_filter_cols_fused(nf::LineArgs<T> a) {
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
    G.nl = G.tl < nvalid ? G.tl + 1 : 0;
    const size_t plane = (size_t)a.realH * (size_t)a.realW;
    const size_t inEs = a.inType == 1 ? 8 : 4, outEs = a.outType == 1 ? 8 : 4;
    G.inP = (const char*)a.in + (size_t)blockIdx.y * plane * inEs;
    G.outP = (char*)a.out + (size_t)blockIdx.y * plane * outEs;
    const void* refP = a.epiZeroRef != nullptr ? (const void*)((const char*)a.epiZeroRef + (size_t)blockIdx.y * plane * outEs) : nullptr;
    const int S = p.nst;
    {
        const nf::StageInfo st = p.st[S - 1];
#define NF_LFF_FIRST(R) lffc_first<T, R>(buf, a, st, G)
        NF_LFF_SWITCH(NF_LFF_FIRST)
#undef NF_LFF_FIRST
    }
    for (int s = S - 2; s >= 1; --s) {
        const nf::StageInfo st = p.st[s];
#define NF_LFF_DIF(R) lff_stage<T, R, true>(buf, a, st, true, G)
        NF_LFF_SWITCH(NF_LFF_DIF)
#undef NF_LFF_DIF
    }
    {
        const nf::StageInfo st = p.st[0];
#define NF_LFF_MID(R) lff_middle<T, R, true>(buf, a, st, G)
        NF_LFF_SWITCH(NF_LFF_MID)
#undef NF_LFF_MID
    }
    for (int s = 1; s <= S - 2; ++s) {
        const nf::StageInfo st = p.st[s];
#define NF_LFF_DIT(R) lff_stage<T, R, true>(buf, a, st, false, G)
        NF_LFF_SWITCH(NF_LFF_DIT)
#undef NF_LFF_DIT
    }
    {
        const nf::StageInfo st = p.st[S - 1];
#define NF_LFF_LAST(R) lffc_last<T, R>(buf, a, st, G, refP)
        NF_LFF_SWITCH(NF_LFF_LAST)
#undef NF_LFF_LAST
    }
}

}  // namespace

namespace nf {

// Plain radices up to 8, two stages or more (the caller has checked): grid.x = row pairs / linesPerWG, grid.y = maps.
template <typename T>
int launch_line_filter_rows_fused(const LineArgs<T>& a, int batch, int threads, size_t lds, hipStream_t stream) {
    static bool attrSet = false;
    if (!attrSet) {
        NF_HIP(hipFuncSetAttribute((const void*)k_line_filter_rows_fused<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        attrSet = true;
    }
    dim3 grid(nf_div_up(a.nLines, a.linesPerWG), batch);
    hipLaunchKernelGGL((k_line_filter_rows_fused<T>), grid, dim3(threads), lds, stream, a);
    return NF_OK;
}

template int launch_line_filter_rows_fused<float>(const LineArgs<float>&, int, int, size_t, hipStream_t);
template int launch_line_filter_rows_fused<double>(const LineArgs<double>&, int, int, size_t, hipStream_t);

// Columns: a.linesPerWG / a.nTiles as Plan2D::axis_filter prepares them for k_line_filter; grid.x = tiles rounded up to the XCDs.
template <typename T>
int launch_line_filter_cols_fused(const LineArgs<T>& a, int batch, int threads, size_t lds, hipStream_t stream) {
    static bool attrSet = false;
    if (!attrSet) {
        NF_HIP(hipFuncSetAttribute((const void*)k_line_filter_cols_fused<T, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        NF_HIP(hipFuncSetAttribute((const void*)k_line_filter_cols_fused<T, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
        attrSet = true;
    }
    dim3 grid(((a.nTiles + 7) / 8) * 8, batch);
    if (threads <= 256)
        hipLaunchKernelGGL((k_line_filter_cols_fused<T, 256>), grid, dim3(threads), lds, stream, a);
    else
        hipLaunchKernelGGL((k_line_filter_cols_fused<T, 1024>), grid, dim3(threads), lds, stream, a);
    return NF_OK;
}

template int launch_line_filter_cols_fused<float>(const LineArgs<float>&, int, int, size_t, hipStream_t);
template int launch_line_filter_cols_fused<double>(const LineArgs<double>&, int, int, size_t, hipStream_t);

}  // namespace nf

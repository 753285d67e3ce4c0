// The matched-filter scales of a tile (nf_matched_filter_scales, include/nemofilt.h): the statements of
// MatchedFilter.buildAndApply (nemo/filters.py:521-777) and of filterMaps' pixel-window undo (:98-101) as a fixed sequence of
// this library's own kernels, enqueued from C++ instead of from the interpreter.  The scales of a tile are independent of each
// other and differ only in their templates, so a GROUP of them (up to nf_mf_tile.scaleBatch) walks the sequence together: every
// stage is one launch with the scales as its batch dimension - a 1-2 Mpix tile then offers the chip eight times the workgroups
// per launch and an eighth of the launches.  Groups are dealt to lanes (HIP streams) that run side by side.  Per scale the
// arithmetic is that of the single-scale entry points, bit for bit (tests/test_gpu_scale_entry.py).
#include "../../include/nemofilt.h"

#include "nf_common.h"
#include "nf_kernels.h"
#include "nf_plan.h"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

// ---- the two scalar steps ------------------------------------------------------------------------------------
// nemo/filters.py:659: signalNorm = y0 / peak (float64 division);  :684: 1.0 / max in float32 arithmetic (numpy >= 2: the
// Python float is weak, the float32 maximum decides the type), widened for storage.
__device__ __forceinline__ double signal_norm(int mode, double src, double y0) {
    if (mode == 0) return __ddiv_rn(y0, src);
    return (double)__fdiv_rn(1.0f, (float)src);
}
__global__ void k_signal_norm(int mode, const double* __restrict__ src, double y0, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = signal_norm(mode, src[0], y0);
}
// one thread per scale of a group: src[b * srcStride] -> the caller's scalar and the group's norms side by side
__global__ void k_signal_norm_group(int mode, const double* __restrict__ src, long long srcStride, double y0, nf::BatchPtrs out,
                                    double* __restrict__ normWs, int nb) {
    const int b = threadIdx.x;
    if (b >= nb || blockIdx.x != 0) return;
    const double v = signal_norm(mode, src[(long long)b * srcStride], y0);
    ((double*)out.p[b])[0] = v;
    normWs[b] = v;
}

// a lane's workspace: every buffer holds B slots (one per scale of a group), slot after slot
struct Layout {
    int B;
    size_t T, F, filtSym, work, sig, planes, pre, zeroRows, noise, noiseBytes, noiseStride, edgeTmp, edgeOut, mapMax, peak, norm, total;
};

int batch_of(const nf_mf_tile& t) { return t.scaleBatch < 1 ? 1 : (t.scaleBatch > NF_SB_MAX ? NF_SB_MAX : t.scaleBatch); }

Layout layout_of(const nf_mf_tile& t) {
    Layout L;
    const size_t npix = (size_t)t.H * (size_t)t.W;
    const size_t N = (size_t)t.N, B = (size_t)batch_of(t);
    const bool yc = t.outputUnits == 0;
    const size_t outB = yc ? 8 : 4;
    size_t off = 0;
    auto take = [&off](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    L.B = (int)B;
    L.T = take(B * N * npix * 4);                    // painted templates, float32
    L.F = take(B * N * npix * 8);                    // their half-stored spectra, complex64
    L.filtSym = take(B * N * npix * 8);              // symmetrised filters, float64
    // transposed intermediate of the inverse transforms: the group's data maps, or one chunk (<= 16 planes) of filtered templates
    size_t work = B * npix * (t.prec == NF_F64 ? 16 : 8);
    const size_t chunk = (B * N < NF_PLANE_SCALES ? B * N : NF_PLANE_SCALES) * npix * 8;
    if (yc && chunk > work) work = chunk;
    L.work = take(work);
    L.sig = take(B * npix * 4);                      // filtered templates
    L.planes = take(yc && N > 1 ? B * N * npix * 4 : 0);
    L.pre = take(B * npix * outB);                   // filtered maps before the pixel-window undo
    L.zeroRows = take(B * (size_t)t.H * 4);           // rows of the filtered maps that hold a zero (for the window undo)
    L.noiseBytes = nf_noise_workspace_bytes(t.H, t.W, t.ny, t.nx, t.numBins);
    L.noiseStride = (L.noiseBytes + 255) & ~(size_t)255;
    L.noise = take(B * L.noiseStride);
    L.edgeTmp = take(t.trimSizePix > 0 ? B * npix : 0);
    L.edgeOut = take(t.trimSizePix > 0 ? B * npix : 0);
    L.mapMax = take(B * NF_MAP_MAX_DOUBLES * 8);
    L.peak = take(B * 8);
    L.norm = take(B * 8);
    L.total = off;
    return L;
}

int check_tile(const nf_mf_tile* t) {
    if (t && t->N > NF_MF_MAXN) {
        nf_set_error("nf_matched_filter_scales: more than NF_MF_MAXN = 6 frequency maps (the limit of the filter build)");
        return NF_ERR_SIZE;
    }
    if (!t || t->H < 1 || t->W < 1 || t->N < 1 || !t->fMaps_d || !t->noiseCov_d || !t->A_d || !t->B_d ||
        !t->C_d || !t->medWeights_d || !t->yEdges_d || !t->xEdges_d || !t->quantiles || t->ny < 1 || t->nx < 1 || t->numBins < 1 ||
        (t->outputUnits != 0 && t->outputUnits != 1) || (t->prec != NF_F32 && t->prec != NF_F64) ||
        ((t->axisWinY_d == nullptr) != (t->axisWinX_d == nullptr))) {
        nf_set_error("nf_matched_filter_scales: bad tile description");
        return NF_ERR_ARG;
    }
    if (t->outputUnits == 0 && (!t->centreYX_d || !t->winY_d || !t->winX_d)) {
        nf_set_error("nf_matched_filter_scales: 'yc' needs the tile centre and the pixel-window tables");
        return NF_ERR_ARG;
    }
    return NF_OK;
}

// ---- lanes: streams and events kept per device for the life of the process ------------------------------------
// All of a device's lane streams and events are made in ONE place, the first time the device asks for its workspace size
// (nf_matched_filter_workspace_bytes, which every caller makes before its first nf_matched_filter_scales) - never in the
// middle of a launch sequence.  A call holds the device's mutex from its fork to its join: the fork / done events are shared
// by all calls on the device, so two host threads enqueueing on one GPU take turns (each call only enqueues: milliseconds).
#define NF_MF_MAXLANES 4            // lane 0 = the caller's stream + 3 library streams (GPU_MAX_HW_QUEUES is 4 on MI355X boxes)
struct Lanes {
    std::mutex mu;                        // held for the whole enqueue of one nf_matched_filter_scales call
    hipStream_t streams[NF_MF_MAXLANES - 1] = {};     // lane l >= 1 runs on streams[l - 1]
    hipEvent_t done[NF_MF_MAXLANES - 1] = {};
    hipEvent_t fork = nullptr;
    bool ready = false;
};
std::mutex g_laneMu;                      // guards the map only
std::map<int, Lanes> g_lanes;

int lanes_for_device(Lanes** out) {
    int dev = 0;
    NF_HIP(hipGetDevice(&dev));
    Lanes* L;
    {
        std::lock_guard<std::mutex> lock(g_laneMu);
        L = &g_lanes[dev];                // (std::map nodes never move: the pointer stays good)
    }
    std::lock_guard<std::mutex> lock(L->mu);
    if (!L->ready) {
        NF_HIP(hipEventCreateWithFlags(&L->fork, hipEventDisableTiming));
        for (int l = 0; l < NF_MF_MAXLANES - 1; ++l) {
            NF_HIP(hipStreamCreateWithFlags(&L->streams[l], hipStreamNonBlocking));
            NF_HIP(hipEventCreateWithFlags(&L->done[l], hipEventDisableTiming));
        }
        L->ready = true;
    }
    *out = L;
    return NF_OK;
}

#define NF_TRY(call)            \
    do {                        \
        int _rc = (call);       \
        if (_rc != NF_OK) return _rc; \
    } while (0)

// the sequence for one group of nb <= L.B scales on one stream
int run_group(nf_plan* plan, const nf_mf_tile& t, const nf_mf_scale* sc, int nb, const Layout& L, char* ws, bool noiseTablesValid,
              hipStream_t st) {
    const int H = t.H, W = t.W, N = t.N;
    const long long npix = (long long)H * W;
    const bool yc = t.outputUnits == 0;
    const int outPrec = yc ? NF_F64 : NF_F32;
    const int outType = yc ? NF_IN_REAL64 : NF_IN_REAL32;
    const size_t es = yc ? 8 : 4;
    float* T = (float*)(ws + L.T);
    char* F = ws + L.F;                                   // complex64 planes [nb][N]
    double* filtSym = (double*)(ws + L.filtSym);
    void* work = ws + L.work;
    float* sig = (float*)(ws + L.sig);
    float* planes = (float*)(ws + L.planes);
    const bool undo = t.axisWinY_d != nullptr;
    int* zeroRows = (int*)(ws + L.zeroRows);
    double* mapMax = (double*)(ws + L.mapMax);
    double* peak = (double*)(ws + L.peak);
    double* normWs = (double*)(ws + L.norm);
    void* S = (void*)st;
    for (int b = 0; b < nb; ++b) {
        if (!sc[b].signalNorm_d || !sc[b].data_d || !sc[b].sn_d || !sc[b].rms_d || !sc[b].mask_d || (yc && N > 1 && !sc[b].relWeights_d)) {
            nf_set_error("nf_matched_filter_scales: a scale lacks an output buffer");
            return NF_ERR_ARG;
        }
        // the group's filtered maps are one batch of the transforms: their buffers lie map after map
        if ((char*)sc[b].data_d != (char*)sc[0].data_d + (size_t)b * (size_t)npix * es) {
            nf_set_error("nf_matched_filter_scales: internal - a group's data maps must lie one after the other");
            return NF_ERR_ARG;
        }
    }
    char* pre = undo ? ws + L.pre : (char*)sc[0].data_d;  // filtered maps [nb] before the window undo
    const double unit = std::sqrt((double)H * (double)W);  // enmap.fft's divisor (the forward transforms are unitary, the inverse ones are not)
    NF_TRY(nf_plan_ensure_prec(plan, NF_F32));
    NF_TRY(nf_plan_ensure_prec(plan, t.prec));
    NF_TRY(nf_plan_ensure_prec(plan, outPrec));

    // unit templates (nemo/filters.py:611-618): the pixel's angular distance is formed once per launch for up to 8 planes (same
    // values as one nf_paint_radial call per plane); then their spectra, all planes of all scales in one transform
    const int nPlanes = nb * N;
    for (int p0 = 0; p0 < nPlanes; p0 += 8) {
        nf::PaintPlanes P;
        memset(&P, 0, sizeof(P));
        P.n = nPlanes - p0 < 8 ? nPlanes - p0 : 8;
        for (int j = 0; j < P.n; ++j) {
            const nf_mf_scale& s1 = sc[(p0 + j) / N];
            const int i = (p0 + j) % N;
            if (!s1.r_d[i] || !s1.prof_d[i] || s1.ntab[i] < 2 || !(s1.dlnr[i] > 0) || !(s1.r0[i] > 0)) {
                nf_set_error("nf_matched_filter_scales: bad radial table");
                return NF_ERR_ARG;
            }
            P.r[j] = s1.r_d[i];
            P.prof[j] = s1.prof_d[i];
            P.slope[j] = s1.slope_d[i];
            P.ntab[j] = s1.ntab[i];
            P.log2r0[j] = (float)(log(s1.r0[i]) / log(2.0));
            P.invDlog2[j] = (float)(1.0 / (s1.dlnr[i] / log(2.0)));
            P.amp[j] = s1.amp[i];
            P.sign[j] = (float)s1.sign[i];
        }
        nf::ProfScope ps(nf::PT_PAINT, st, (double)npix * 4.0 * P.n);
        if (P.n == 1) {
            // (one plane: the single-table kernel is the shorter one - 52 against 61 us at 4096^2; same values)
            const nf_mf_scale& s1 = sc[p0 / N];
            const int i = p0 % N;
            NF_TRY(nf::paint_radial(T + (size_t)p0 * npix, H, W, t.A_d, t.B_d, t.C_d, s1.r_d[i], s1.prof_d[i], s1.slope_d[i], s1.ntab[i],
                                    s1.r0[i], s1.dlnr[i], s1.amp[i], s1.sign[i], st));
        } else {
            NF_TRY(nf::paint_radial_planes(T + (size_t)p0 * npix, H, W, t.A_d, t.B_d, t.C_d, P, st));
        }
    }
    // One frequency map: the column transform of the template's spectrum, the filter build and the filtered template's column
    // transform walk the same column tiles - one kernel does all three and the spectrum never reaches HBM (k_template_cols;
    // per element the arithmetic of the three, bit for bit).  Otherwise: transform, then build, then transform back.
    const bool fuse = N == 1 && plan->pf.template_filter_ok(nb);
    if (fuse) {
        NF_TRY(plan->pf.forward(T, NF_IN_REAL32, F, nPlanes, nullptr, nullptr, 1, unit, st, 1, 1));      // row pass only
    } else {
        NF_TRY(nf_fft2_forward_half(plan, NF_F32, T, NF_IN_REAL32, F, nPlanes, nullptr, nullptr, 1, S));
    }
    // the filters (:621-627) and their symmetrised forms for the half-spectrum inverse
    if (!fuse) {
        nf::WeightVec wv;
        memset(&wv, 0, sizeof(wv));
        for (int i = 0; i < N; ++i) wv.v[i] = t.w[i];
        nf::BatchPtrs fp;
        memset(&fp, 0, sizeof(fp));
        int nCubes = 0;
        for (int b = 0; b < nb; ++b) {
            fp.p[b] = sc[b].filt_d;
            nCubes += sc[b].filt_d != nullptr;
        }
        // (the covariance of a mode is read once for all entries where the build shares its inverse, once per entry otherwise)
        const int covReads = nf::build_filter_shares_inverse(nb, N) ? 1 : nb;
        nf::ProfScope ps(nf::PT_BUILD_FILTER, st,
                         (double)npix * (covReads * (double)N * N * (t.covType == NF_F32 ? 4.0 : 8.0) + nb * (N * 4.0 + N * 4.0) + nCubes * N * 4.0));
        NF_TRY(nf::build_filter_sym_batch(t.covType, NF_F32, t.noiseCov_d, F, wv, fp, nb, filtSym, N, H, W, st));
    }
    nf::BatchPtrs normOut, relOut;
    memset(&normOut, 0, sizeof(normOut));
    memset(&relOut, 0, sizeof(relOut));
    for (int b = 0; b < nb; ++b) {
        normOut.p[b] = sc[b].signalNorm_d;
        relOut.p[b] = sc[b].relWeights_d;
    }
    nf::TemplateFuse tf;
    memset(&tf, 0, sizeof(tf));
    if (fuse) {
        tf.cov = t.noiseCov_d;
        tf.w0 = t.w[0];
        tf.sym = filtSym;
        for (int b = 0; b < nb; ++b) tf.filt[b] = sc[b].filt_d;
    }
    // normalisation
    if (yc) {
        // the y0-scaled templates (:639-646) are the unit templates times templateScale: the factor rides on the load.  Up to 8
        // planes (the room in the fused kernel's tables) per transform with one map, 16 (the factor table) with several.
        const int per = N == 1 ? 8 : NF_PLANE_SCALES / N;               // scales per transform
        if (fuse) {
            double planeScale[8];
            for (int j = 0; j < nb; ++j) planeScale[j] = sc[j].templateScale[0];
            NF_TRY(plan->pf.template_filter(F, nb, tf, t.covType, planeScale, t.winY_d, t.winX_d, work, sig, NF_IN_REAL32, t.centreRowLo,
                                            t.centreRowHi, 1, unit, st));
        }
        for (int b0 = 0; b0 < nb && !fuse; b0 += per) {
            const int cnt = nb - b0 < per ? nb - b0 : per;
            double planeScale[NF_PLANE_SCALES];
            for (int j = 0; j < cnt; ++j)
                for (int i = 0; i < N; ++i) planeScale[j * N + i] = sc[b0 + j].templateScale[i];
            const char* Fc = F + (size_t)b0 * N * npix * 8;
            const double* fs = filtSym + (size_t)b0 * N * npix;
            if (N == 1) {
                // one map: only its rows around the tile centre are read below, so only those are computed
                NF_TRY(plan->pf.inverse_half(Fc, work, sig + (size_t)b0 * npix, NF_IN_REAL32, cnt, 0, unit, st, fs, 1, npix, t.winY_d, t.winX_d,
                                             nullptr, nullptr, nullptr, planeScale, t.centreRowLo, t.centreRowHi));
            } else {
                // several maps: the per-frequency filtered templates are wanted anyway (:662-667); their sum in real space, in
                // plane order, is the filtered template (:647 -> :847 sums after the inverse transforms)
                NF_TRY(plan->pf.inverse_half(Fc, work, planes + (size_t)b0 * N * npix, NF_IN_REAL32, cnt * N, 0, unit, st, fs, 1, npix,
                                             t.winY_d, t.winX_d, nullptr, nullptr, nullptr, planeScale, 0, 0));
            }
        }
        if (N > 1) {
            // (the sum pass also leaves the per-workgroup maxima of the summed map: the first half of nf_map_max)
            nf::ProfScope ps(nf::PT_APPLY_FILTER, st, (double)npix * (N + 1) * 4.0 * nb);
            NF_TRY(nf::sum_planes_max(planes, sig, N, npix, mapMax, st, nb));
        }
        {
            // the filtered templates at the tile centre (:654-658), one job per scale
            nf::SplineBatch sb;
            memset(&sb, 0, sizeof(sb));
            for (int b = 0; b < nb; ++b) {
                sb.map[b] = sig + (size_t)b * npix;
                sb.yx[b] = t.centreYX_d;
                sb.out[b] = peak + b;
                sb.n[b] = 1;
                // class 1: the point is known to be interior (an edge list of no entries); else it is offered to both kernels
                sb.edge[b] = t.centreClass == 1 ? (const int*)t.centreYX_d : nullptr;
                sb.nEdge[b] = t.centreClass == 1 ? 0 : 1;
            }
            nf::ProfScope ps(nf::PT_SPLINE, st, 0.0);
            NF_TRY(nf::spline_eval_batch(NF_F32, sb, nb, H, W, t.halfWin, st));
        }
        if (N > 1) {
            // one finishing kernel: the maximum and its position (:661), the N planes there (:662-667), signalNorm (:659)
            nf::ProfScope ps(nf::PT_MAP_MAX, st, 0.0);
            NF_TRY(nf::map_max_finish_weights_batch(mapMax, npix, planes, N, relOut, peak, t.y0, normOut, normWs, nb, st));
        } else {
            hipLaunchKernelGGL(k_signal_norm_group, dim3(1), dim3(64), 0, st, 0, peak, 1LL, t.y0, normOut, normWs, nb);
            NF_CHECK_LAUNCH("k_signal_norm_group");
        }
    } else {
        if (fuse)
            NF_TRY(plan->pf.template_filter(F, nb, tf, t.covType, nullptr, nullptr, nullptr, work, sig, NF_IN_REAL32, 0, 0, 1, unit, st));
        else
            NF_TRY(plan->pf.inverse_half(F, work, sig, NF_IN_REAL32, nb, 0, unit, st, filtSym, N, (long long)N * npix, nullptr, nullptr,
                                         nullptr, nullptr, nullptr, nullptr, 0, 0, (long long)N * npix));
        {
            nf::ProfScope ps(nf::PT_MAP_MAX, st, (double)npix * 4.0 * nb);
            NF_TRY(nf::map_max(NF_F32, sig, npix, mapMax, st, nb, npix));
        }
        hipLaunchKernelGGL(k_signal_norm_group, dim3(1), dim3(64), 0, st, 1, mapMax, (long long)NF_MAP_MAX_DOUBLES, 0.0, normOut, normWs, nb);
        NF_CHECK_LAUNCH("k_signal_norm_group");
    }
    // apply (:696, :714): every scale's filter on the same data spectra (batch stride 0 on the input), signalNorm and the
    // point-source mask at the store
    if (t.prec == NF_F32)
        NF_TRY(plan->pf.inverse_half(t.fMaps_d, work, pre, outType, nb, 0, unit, st, filtSym, N, (long long)N * npix, nullptr, nullptr, normWs,
                                     t.psMask_d, nullptr, nullptr, 0, 0, 0, 1));
    else
        NF_TRY(plan->pd.inverse_half(t.fMaps_d, work, pre, outType, nb, 0, unit, st, filtSym, N, (long long)N * npix, nullptr, nullptr, normWs,
                                     t.psMask_d, nullptr, nullptr, 0, 0, 0, 1));
    // noise maps (:716-722), made in their output buffers and masked there in place
    {
        nf::NoiseGrid g;
        memset(&g, 0, sizeof(g));
        g.ny = t.ny;
        g.nx = t.nx;
        g.overlap = t.overlap;
        g.numBins = t.numBins;
        g.estimator = t.estimator;
        g.pctQ = 68.3 / 100.0;
        g.yEdges = t.yEdges_d;
        g.xEdges = t.xEdges_d;
        if (t.numBins < 1 || t.numBins > 16 || t.overlap < 0) {
            nf_set_error("nf_matched_filter_scales: bad noise grid");
            return NF_ERR_ARG;
        }
        for (int i = 0; i < t.numBins; ++i) g.quantiles[i] = t.quantiles[i];
        float* rmsList[NF_SB_MAX];
        for (int b = 0; b < nb; ++b) rmsList[b] = sc[b].rms_d;
        nf::ProfScope ps(nf::PT_NOISE_MAP, st,
                         (double)npix * nb * ((double)es + (noiseTablesValid ? 0.0 : (t.wtype == NF_F32 ? 4.0 : 8.0)) + 4.0));
        if (nb == 1 || nf::noise_batchable(g)) {
            NF_TRY(nf::noise_map_batch(outPrec, pre, (long long)npix * (long long)es, t.medWeights_d, t.wtype, rmsList, nb, L.B, H, W, g,
                                       ws + L.noise, L.noiseBytes, (long long)L.noiseStride, noiseTablesValid, st));
        } else {
            for (int b = 0; b < nb; ++b)
                NF_TRY(nf::noise_map(outPrec, pre + (size_t)b * npix * es, t.medWeights_d, t.wtype, rmsList[b], H, W, g,
                                     ws + L.noise + (size_t)b * L.noiseStride, L.noiseBytes, false, st));
        }
    }
    // S/N and masks (:725-756)
    nf::FinalizeBatch fb;
    memset(&fb, 0, sizeof(fb));
    for (int b = 0; b < nb; ++b) {
        unsigned char* edge = nullptr;
        if (t.trimSizePix > 0) {
            edge = (unsigned char*)(ws + L.edgeOut) + (size_t)b * npix;
            NF_TRY(nf_edge_check(outPrec, pre + (size_t)b * npix * es, t.psMask_d, (unsigned char*)(ws + L.edgeTmp) + (size_t)b * npix, edge, H,
                                 W, t.trimSizePix, S));
        }
        fb.filtered[b] = pre + (size_t)b * npix * es;
        fb.rms[b] = sc[b].rms_d;
        fb.edgeCheck[b] = edge;
        fb.snOut[b] = sc[b].sn_d;
        fb.rmsOut[b] = sc[b].rms_d;
        fb.maskOut[b] = sc[b].mask_d;
        fb.chunkRange[b] = sc[b].snChunkRange_d;
        fb.zeroRows[b] = undo ? zeroRows + (size_t)b * H : nullptr;
    }
    if (undo) NF_HIP(hipMemsetAsync(zeroRows, 0, (size_t)nb * H * sizeof(int), st));
    {
        nf::ProfScope ps(nf::PT_FINALIZE, st, (double)npix * nb * (2.0 * es + 4.0 + 3.0 + 1.0));
        NF_TRY(nf::finalize_batch(outPrec, fb, nb, t.surveyMask_d, t.psMask_d, H, W, t.apodPix, 1, st));
    }
    // pixel-window undo (:98-101): columns, then rows with the zeros of the input restored at the store
    if (undo) {
        const int ty = yc ? 1 : 2;
        if (yc) {
            NF_TRY(plan->pd.axis_filter(1, pre, ty, sc[0].data_d, ty, nb, t.axisWinY_d, nullptr, st, nullptr));
            NF_TRY(plan->pd.axis_filter(0, sc[0].data_d, ty, sc[0].data_d, ty, nb, t.axisWinX_d, pre, st, zeroRows));
        } else {
            NF_TRY(plan->pf.axis_filter(1, pre, ty, sc[0].data_d, ty, nb, t.axisWinY_d, nullptr, st, nullptr));
            NF_TRY(plan->pf.axis_filter(0, sc[0].data_d, ty, sc[0].data_d, ty, nb, t.axisWinX_d, pre, st, zeroRows));
        }
    }
    return NF_OK;
}

}  // namespace

extern "C" {

size_t nf_matched_filter_workspace_bytes(const nf_mf_tile* tile) {
    if (check_tile(tile) != NF_OK) return 0;
    Lanes* lanes = nullptr;
    if (lanes_for_device(&lanes) != NF_OK) return 0;      // the device's lane streams and events exist from here on
    return layout_of(*tile).total;
}

int nf_matched_filter_max_lanes(void) { return NF_MF_MAXLANES; }

int nf_signal_norm(int outputUnits, const double* src_d, double y0, double* out_d, void* stream) {
    if (!src_d || !out_d || (outputUnits != 0 && outputUnits != 1)) {
        nf_set_error("nf_signal_norm: bad arguments");
        return NF_ERR_ARG;
    }
    hipLaunchKernelGGL(k_signal_norm, dim3(1), dim3(64), 0, (hipStream_t)stream, outputUnits, src_d, y0, out_d);
    NF_CHECK_LAUNCH("k_signal_norm");
    return NF_OK;
}

int nf_matched_filter_scales(nf_plan* plan, const nf_mf_tile* tile, const nf_mf_scale* scales, int nScales, void* const* work_d,
                             int nLanes, void* stream) {
    NF_TRY(check_tile(tile));
    if (!plan || !scales || nScales < 0 || !work_d || nLanes < 1) {
        nf_set_error("nf_matched_filter_scales: bad arguments");
        return NF_ERR_ARG;
    }
    if (nScales == 0) return NF_OK;
    if (nLanes > nScales) nLanes = nScales;
    if (nLanes > NF_MF_MAXLANES) nLanes = NF_MF_MAXLANES;
    for (int l = 0; l < nLanes; ++l)
        if (!work_d[l] || ((size_t)work_d[l] & 255)) {
            nf_set_error("nf_matched_filter_scales: a lane's workspace is missing or not 256-byte aligned");
            return NF_ERR_ARG;
        }
    const Layout L = layout_of(*tile);
    hipStream_t mainStream = (hipStream_t)stream;
    Lanes* lanes = nullptr;
    NF_TRY(lanes_for_device(&lanes));
    // groups of scales: as many per group as the workspaces have slots, but no fewer groups than lanes
    int per = (nScales + nLanes - 1) / nLanes;
    if (per > L.B) per = L.B;
    const size_t es = tile->outputUnits == 0 ? 8 : 4, mapBytes = (size_t)tile->H * (size_t)tile->W * es;
    // from the fork to the join nobody else records or waits on this device's lane events
    std::lock_guard<std::mutex> enqueue(lanes->mu);
    if (nLanes > 1) {
        NF_HIP(hipEventRecord(lanes->fork, mainStream));
        for (int l = 1; l < nLanes; ++l) NF_HIP(hipStreamWaitEvent(lanes->streams[l - 1], lanes->fork, 0));
    }
    int rc = NF_OK;
    int grp = 0;
    for (int k = 0; k < nScales && rc == NF_OK; ++grp) {
        // (a group's filtered maps are one batch of the transforms: it ends where the callers' buffers stop lying map after map)
        int nb = 1;
        while (nb < per && k + nb < nScales && scales[k + nb].data_d && scales[k].data_d &&
               (char*)scales[k + nb].data_d == (char*)scales[k].data_d + (size_t)nb * mapBytes)
            ++nb;
        const int l = grp % nLanes;
        // the noise block tables of a lane's workspace: valid after the lane's first group, or when the caller says so
        const bool tables = grp >= nLanes || (l < 32 && ((tile->noiseTablesLanes >> l) & 1u));
        rc = run_group(plan, *tile, scales + k, nb, L, (char*)work_d[l], tables, l == 0 ? mainStream : lanes->streams[l - 1]);
        k += nb;
    }
    // join whatever was enqueued, also after an error: the caller's stream must not run ahead of the lanes
    for (int l = 1; l < nLanes; ++l) {
        hipError_t e = hipEventRecord(lanes->done[l - 1], lanes->streams[l - 1]);
        if (e == hipSuccess) e = hipStreamWaitEvent(mainStream, lanes->done[l - 1], 0);
        if (e != hipSuccess && rc == NF_OK) rc = nf_hip_fail(e, "nf_matched_filter_scales: join");
    }
    return rc;
}

}  // extern "C"

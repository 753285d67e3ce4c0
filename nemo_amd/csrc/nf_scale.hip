// One matched-filter scale per call (nf_matched_filter_scales, include/nemofilt.h): the statements of
// MatchedFilter.buildAndApply (nemo/filters.py:521-777) and of filterMaps' pixel-window undo (:98-101) as a fixed sequence of
// this library's own entry points, enqueued from C++ instead of from the interpreter.  No kernel of its own beyond the
// scalar step between them (signalNorm); scales of a tile are independent of each other, so
// they are dealt to lanes (HIP streams) that run side by side - small tiles do not fill the chip one kernel at a time.
#include "../../include/nemofilt.h"

#include "nf_common.h"
#include "nf_kernels.h"

#include <map>
#include <mutex>
#include <vector>

namespace {

// ---- the two scalar steps ------------------------------------------------------------------------------------
// nemo/filters.py:659: signalNorm = y0 / peak (float64 division);  :684: 1.0 / max in float32 arithmetic (numpy >= 2: the
// Python float is weak, the float32 maximum decides the type), widened for storage.
__global__ void k_signal_norm(int mode, const double* __restrict__ src, double y0, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 0) {
        out[0] = __ddiv_rn(y0, src[0]);
    } else {
        const float m = (float)src[0];
        out[0] = (double)__fdiv_rn(1.0f, m);
    }
}

struct Layout {
    size_t T, F, filtSym, work, sig, planes, pre, zeroRows, noise, noiseBytes, edgeTmp, edgeOut, mapMax, peak, total;
};

Layout layout_of(const nf_mf_tile& t) {
    Layout L;
    const size_t npix = (size_t)t.H * (size_t)t.W;
    const size_t N = (size_t)t.N;
    const bool yc = t.outputUnits == 0;
    const size_t nT = N;
    const size_t outB = yc ? 8 : 4;
    size_t off = 0;
    auto take = [&off](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    L.T = take(nT * npix * 4);                       // painted templates, float32
    L.F = take(nT * npix * 8);                       // their half-stored spectra, complex64
    L.filtSym = take(N * npix * 8);                  // symmetrised filter, float64
    size_t work = npix * (t.prec == NF_F64 ? 16 : 8);          // transposed intermediate of the inverse transforms
    if (yc && N > 1 && N * npix * 8 > work) work = N * npix * 8;
    L.work = take(work);
    L.sig = take(npix * 4);                          // filtered template
    L.planes = take(yc && N > 1 ? N * npix * 4 : 0);
    L.pre = take(npix * outB);                       // filtered map before the pixel-window undo
    L.zeroRows = take((size_t)t.H * 4);               // rows of the filtered map that hold a zero (for the window undo)
    L.noiseBytes = nf_noise_workspace_bytes(t.H, t.W, t.ny, t.nx, t.numBins);
    L.noise = take(L.noiseBytes);
    L.edgeTmp = take(t.trimSizePix > 0 ? npix : 0);
    L.edgeOut = take(t.trimSizePix > 0 ? npix : 0);
    L.mapMax = take(NF_MAP_MAX_DOUBLES * 8);
    L.peak = take(8);
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

// the sequence for one scale on one stream
int run_scale(nf_plan* plan, const nf_mf_tile& t, const nf_mf_scale& sc, const Layout& L, char* ws, bool noiseTablesValid, hipStream_t st) {
    const int H = t.H, W = t.W, N = t.N;
    const long long npix = (long long)H * W;
    const bool yc = t.outputUnits == 0;
    const int outPrec = yc ? NF_F64 : NF_F32;
    const int outType = yc ? NF_IN_REAL64 : NF_IN_REAL32;
    float* T = (float*)(ws + L.T);
    char* F = ws + L.F;                                   // complex64 planes
    double* filtSym = (double*)(ws + L.filtSym);
    void* work = ws + L.work;
    float* sig = (float*)(ws + L.sig);
    float* planes = (float*)(ws + L.planes);
    const bool undo = t.axisWinY_d != nullptr;
    void* pre = undo ? (void*)(ws + L.pre) : sc.data_d;
    float* rms0 = sc.rms_d;                                // the noise map is made in its output buffer, masked there in place
    int* zeroRows = (int*)(ws + L.zeroRows);
    double* mapMax = (double*)(ws + L.mapMax);
    double* peak = (double*)(ws + L.peak);
    void* S = (void*)st;
    if (!sc.signalNorm_d || !sc.data_d || !sc.sn_d || !sc.rms_d || !sc.mask_d || (yc && N > 1 && !sc.relWeights_d)) {
        nf_set_error("nf_matched_filter_scales: a scale lacks an output buffer");
        return NF_ERR_ARG;
    }
    // unit templates (nemo/filters.py:611-618) and their spectra, all planes in one transform
    if (N == 1) {
        NF_TRY(nf_paint_radial(T, H, W, t.A_d, t.B_d, t.C_d, sc.r_d[0], sc.prof_d[0], sc.slope_d[0], sc.ntab[0], sc.r0[0], sc.dlnr[0],
                               sc.amp[0], sc.sign[0], S));
    } else {
        // all frequencies in one launch: the pixel's angular distance is formed once (same values as N nf_paint_radial calls)
        nf::PaintPlanes P;
        P.n = N;
        for (int i = 0; i < N; ++i) {
            if (!sc.r_d[i] || !sc.prof_d[i] || sc.ntab[i] < 2 || !(sc.dlnr[i] > 0) || !(sc.r0[i] > 0)) {
                nf_set_error("nf_matched_filter_scales: bad radial table");
                return NF_ERR_ARG;
            }
            P.r[i] = sc.r_d[i];
            P.prof[i] = sc.prof_d[i];
            P.slope[i] = sc.slope_d[i];
            P.ntab[i] = sc.ntab[i];
            P.log2r0[i] = (float)(log(sc.r0[i]) / log(2.0));
            P.invDlog2[i] = (float)(1.0 / (sc.dlnr[i] / log(2.0)));
            P.amp[i] = sc.amp[i];
            P.sign[i] = (float)sc.sign[i];
        }
        nf::ProfScope ps(nf::PT_PAINT, st, (double)npix * 4.0 * N);
        NF_TRY(nf::paint_radial_planes(T, H, W, t.A_d, t.B_d, t.C_d, P, st));
    }
    NF_TRY(nf_fft2_forward_half(plan, NF_F32, T, NF_IN_REAL32, F, N, nullptr, nullptr, 1, S));
    // the filter (:621-627) and its symmetrised form for the half-spectrum inverse
    NF_TRY(nf_build_filter_sym_half(t.covType, NF_F32, t.noiseCov_d, F, t.w, sc.filt_d, filtSym, N, H, W, S));
    // normalisation
    if (yc) {
        // the y0-scaled templates (:639-646) are the unit templates times templateScale: the factor rides on the load
        if (N == 1) {
            // one map: only its rows around the tile centre are read below, so only those are computed
            NF_TRY(nf_fft2_inverse_half(plan, NF_F32, F, filtSym, 1, 0, t.winY_d, t.winX_d, sc.templateScale, work, sig, NF_IN_REAL32, 1,
                                        t.centreRowLo, t.centreRowHi, 0, S));
        } else {
            // several maps: the per-frequency filtered templates are wanted anyway (:662-667); their sum in real space, in
            // plane order, is the filtered template (:647 -> :847 sums after the inverse transforms)
            NF_TRY(nf_fft2_inverse_half(plan, NF_F32, F, filtSym, 1, 1, t.winY_d, t.winX_d, sc.templateScale, work, planes,
                                        NF_IN_REAL32, N, 0, 0, 0, S));
            // (the sum pass also leaves the per-workgroup maxima of the summed map: the first half of nf_map_max)
            nf::ProfScope ps(nf::PT_APPLY_FILTER, st, (double)npix * (N + 1) * 4.0);
            NF_TRY(nf::sum_planes_max(planes, sig, N, npix, mapMax, st));
        }
        NF_TRY(nf_spline_eval(NF_F32, sig, 0, H, W, t.centreYX_d, 1, t.halfWin, t.centreClass, nullptr, 0, peak, S));
        if (N > 1) {
            // one finishing kernel: the maximum and its position (:661), the N planes there (:662-667), signalNorm (:659)
            nf::ProfScope ps(nf::PT_MAP_MAX, st, 0.0);
            NF_TRY(nf::map_max_finish_weights(mapMax, npix, planes, N, sc.relWeights_d, peak, t.y0, sc.signalNorm_d, st));
        } else {
            hipLaunchKernelGGL(k_signal_norm, dim3(1), dim3(64), 0, st, 0, peak, t.y0, sc.signalNorm_d);
            NF_CHECK_LAUNCH("k_signal_norm");
        }
    } else {
        NF_TRY(nf_fft2_inverse_half(plan, NF_F32, F, filtSym, N, 0, nullptr, nullptr, nullptr, work, sig, NF_IN_REAL32, 1, 0, 0, 0, S));
        NF_TRY(nf_map_max(NF_F32, sig, npix, mapMax, S));
        hipLaunchKernelGGL(k_signal_norm, dim3(1), dim3(64), 0, st, 1, mapMax, 0.0, sc.signalNorm_d);
        NF_CHECK_LAUNCH("k_signal_norm");
    }
    // apply (:696, :714), noise map, S/N and masks (:716-756)
    NF_TRY(nf_fft2_inverse_half_epi(plan, t.prec, t.fMaps_d, filtSym, N, nullptr, nullptr, work, pre, outType, 0, sc.signalNorm_d,
                                    t.psMask_d, nullptr, S));
    NF_TRY(nf_noise_map(outPrec, pre, t.medWeights_d, t.wtype, rms0, H, W, t.ny, t.nx, t.yEdges_d, t.xEdges_d, t.overlap, t.numBins,
                        t.quantiles, t.estimator, ws + L.noise, L.noiseBytes, noiseTablesValid ? 1 : 0, S));
    unsigned char* edge = nullptr;
    if (t.trimSizePix > 0) {
        edge = (unsigned char*)(ws + L.edgeOut);
        NF_TRY(nf_edge_check(outPrec, pre, t.psMask_d, (unsigned char*)(ws + L.edgeTmp), edge, H, W, t.trimSizePix, S));
    }
    NF_TRY(nf_finalize(outPrec, pre, rms0, t.surveyMask_d, t.psMask_d, edge, sc.sn_d, sc.rms_d, sc.mask_d, H, W, t.apodPix, 1, sc.snChunkRange_d,
                       undo ? zeroRows : nullptr, S));
    // pixel-window undo (:98-101): columns, then rows with the zeros of the input restored at the store
    if (undo) {
        const int ty = yc ? 1 : 2;
        NF_TRY(nf_axis_filter(plan, outPrec, 1, pre, ty, sc.data_d, ty, 1, t.axisWinY_d, nullptr, nullptr, S));
        NF_TRY(nf_axis_filter(plan, outPrec, 0, sc.data_d, ty, sc.data_d, ty, 1, t.axisWinX_d, pre, zeroRows, S));
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
    // from the fork to the join nobody else records or waits on this device's lane events
    std::lock_guard<std::mutex> enqueue(lanes->mu);
    if (nLanes > 1) {
        NF_HIP(hipEventRecord(lanes->fork, mainStream));
        for (int l = 1; l < nLanes; ++l) NF_HIP(hipStreamWaitEvent(lanes->streams[l - 1], lanes->fork, 0));
    }
    int rc = NF_OK;
    for (int k = 0; k < nScales && rc == NF_OK; ++k) {
        const int l = k % nLanes;
        // the noise block tables of a lane's workspace: valid after the lane's first scale, or when the caller says so
        const bool tables = k >= nLanes || (l < 32 && ((tile->noiseTablesLanes >> l) & 1u));
        rc = run_scale(plan, *tile, scales[k], L, (char*)work_d[l], tables, l == 0 ? mainStream : lanes->streams[l - 1]);
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

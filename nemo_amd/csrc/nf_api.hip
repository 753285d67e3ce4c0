// extern "C" boundary of libnemofilt (declarations: include/nemofilt.h).
#include "../../include/nemofilt.h"

#include "nf_common.h"
#include "nf_fft.h"
#include "nf_kernels.h"
#include "nf_plan.h"

#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

static thread_local std::string g_lastError;

void nf_set_error(const std::string& msg) { g_lastError = msg; }

int nf_hip_fail(hipError_t e, const char* what) {
    g_lastError = std::string(what) + ": " + hipGetErrorString(e);
    return NF_ERR_HIP;
}

// ---- profiler: event pairs around entry points, resolved when read --------------------------------------------
namespace {
struct ProfRec {
    int tag;
    double bytes;
    hipEvent_t a, b;
};
const int kProfMax = 1 << 16;
bool g_profOn = false;
unsigned long long g_profMask = ~0ULL;      // entry points (tags) that record events while profiling is on
std::mutex g_profMu;
std::vector<ProfRec> g_profRecs;       // used records [0, g_profUsed)
int g_profUsed = 0;
const char* kProfNames[nf::PT_COUNT] = {
    "fft_rows_f64", "fft_cols_f64", "fft_rows_f32", "fft_cols_f32", "cross_power", "gaussian_filter", "build_filter",
    "apply_filter", "window_mul", "paint_radial", "scale_mask", "noise_map", "median_stack", "edge_check", "finalize",
    "restore_zeros", "subtract", "convolve2d", "label_components", "object_stats", "map_max", "ring_flags", "spline_eval",
    "axis_filter_rows", "axis_filter_cols"};
}  // namespace

nf::ProfScope::ProfScope(int tag, hipStream_t s, double bytes) : slot(-1), stream(s) {
    if (!g_profOn || !((g_profMask >> tag) & 1ULL)) return;
    std::lock_guard<std::mutex> lock(g_profMu);
    if (g_profUsed >= kProfMax) return;
    if (g_profUsed >= (int)g_profRecs.size()) {
        ProfRec r;
        r.tag = 0;
        r.bytes = 0;
        if (hipEventCreate(&r.a) != hipSuccess) return;
        if (hipEventCreate(&r.b) != hipSuccess) {
            (void)hipEventDestroy(r.a);
            return;
        }
        g_profRecs.push_back(r);
    }
    slot = g_profUsed++;
    g_profRecs[slot].tag = tag;
    g_profRecs[slot].bytes = bytes;
    (void)hipEventRecord(g_profRecs[slot].a, s);
}

nf::ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_profMu);
    (void)hipEventRecord(g_profRecs[slot].b, stream);
}

int nf_plan_ensure_prec(nf_plan* p, int prec) {
    std::lock_guard<std::mutex> lock(p->mu);
    if (prec == NF_F32) {
        if (!p->haveF) {
            int rc = p->pf.init(p->H, p->W);
            if (rc) return rc;
            p->haveF = true;
        }
    } else if (prec == NF_F64) {
        if (!p->haveD) {
            int rc = p->pd.init(p->H, p->W);
            if (rc) return rc;
            p->haveD = true;
        }
    } else {
        nf_set_error("precision must be NF_F32 or NF_F64");
        return NF_ERR_ARG;
    }
    return NF_OK;
}

namespace {

inline int ensure_prec(nf_plan* p, int prec) { return nf_plan_ensure_prec(p, prec); }

inline hipStream_t S(void* s) { return (hipStream_t)s; }

}  // namespace

extern "C" {

int nf_version(void) { return 1; }

const char* nf_last_error(void) { return g_lastError.c_str(); }

int nf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return nf_hip_fail(e, "hipGetDeviceCount");
    return n;
}

int nf_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_profMu);
    g_profOn = on != 0;
    g_profUsed = 0;
    return NF_OK;
}

int nf_profile_select(unsigned long long tagMask) {
    std::lock_guard<std::mutex> lock(g_profMu);
    g_profMask = tagMask;
    return NF_OK;
}

int nf_profile_num_tags(void) { return nf::PT_COUNT; }

const char* nf_profile_tag_name(int tag) { return (tag >= 0 && tag < nf::PT_COUNT) ? kProfNames[tag] : ""; }

int nf_profile_read(int tag, long long* count, double* totalMs, double* totalBytes) {
    if (tag < 0 || tag >= nf::PT_COUNT || !count || !totalMs || !totalBytes) {
        nf_set_error("nf_profile_read: bad arguments");
        return NF_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_profMu);
    long long n = 0;
    double ms = 0, by = 0;
    for (int i = 0; i < g_profUsed; ++i) {
        if (g_profRecs[i].tag != tag) continue;
        NF_HIP(hipEventSynchronize(g_profRecs[i].b));
        float t = 0;
        NF_HIP(hipEventElapsedTime(&t, g_profRecs[i].a, g_profRecs[i].b));
        ms += t;
        by += g_profRecs[i].bytes;
        ++n;
    }
    *count = n;
    *totalMs = ms;
    *totalBytes = by;
    return NF_OK;
}

int nf_plan_create(int H, int W, nf_plan** plan) {
    if (!plan || H < 1 || W < 1) {
        nf_set_error("nf_plan_create: bad arguments");
        return NF_ERR_ARG;
    }
    nf_plan* p = new (std::nothrow) nf_plan();
    if (!p) return NF_ERR_NOMEM;
    p->H = H;
    p->W = W;
    p->haveF = p->haveD = false;
    // (new nf_plan() value-initialises both Plan2D members: numbers and pointers zero, the host tables empty)
    // set-up that must not happen inside somebody's launch sequence or graph capture: the per-device address of the static
    // scratch array of nf_xmatch_self_static is resolved here, with this device current (it may load the code object)
    void* scratch = nullptr;
    int rc = nf::xmatch_static_scratch(&scratch);
    if (rc != NF_OK) {
        delete p;
        return rc;
    }
    *plan = p;
    return NF_OK;
}

int nf_plan_destroy(nf_plan* p) {
    if (!p) return NF_OK;
    if (p->haveF) p->pf.destroy();
    if (p->haveD) p->pd.destroy();
    delete p;
    return NF_OK;
}

int nf_plan_digit_reversal(nf_plan* p, int prec, int axis, unsigned short* rev, int* bluestein) {
    if (!p || !rev || !bluestein || (axis != 0 && axis != 1)) {
        nf_set_error("nf_plan_digit_reversal: bad arguments");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    const std::vector<unsigned short>& v = prec == NF_F32 ? (axis == 0 ? p->pf.revRow : p->pf.revCol) : (axis == 0 ? p->pd.revRow : p->pd.revCol);
    const int n = axis == 0 ? p->W : p->H;
    *bluestein = prec == NF_F32 ? (axis == 0 ? p->pf.row.bluestein : p->pf.col.bluestein) : (axis == 0 ? p->pd.row.bluestein : p->pd.col.bluestein);
    if (*bluestein) return NF_OK;      // (the table then belongs to the padded length: not handed out)
    if ((int)v.size() < n) {
        nf_set_error("nf_plan_digit_reversal: plan table missing");
        return NF_ERR_ARG;
    }
    for (int i = 0; i < n; ++i) rev[i] = v[i];
    return NF_OK;
}

int nf_axis_filter(nf_plan* p, int prec, int axis, const void* in, int inType, void* out, int outType, int batch,
                   const double* wperm, const void* zeroRef, const int* zeroRows, void* stream) {
    if (!p || !in || !out || !wperm || batch < 1 || (axis != 0 && axis != 1) || (inType != NF_IN_REAL64 && inType != NF_IN_REAL32) ||
        (outType != NF_IN_REAL64 && outType != NF_IN_REAL32)) {
        nf_set_error("nf_axis_filter: bad arguments (real maps in and out)");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    if (prec == NF_F32) return p->pf.axis_filter(axis, in, inType, out, outType, batch, wperm, zeroRef, S(stream), zeroRows);
    return p->pd.axis_filter(axis, in, inType, out, outType, batch, wperm, zeroRef, S(stream), zeroRows);
}

int nf_fft2_forward(nf_plan* p, int prec, const void* in, int inType, void* out, int batch, const double* apodY,
                    const double* apodX, int unitary, void* stream) {
    if (!p || !in || !out || batch < 1 || inType < 0 || inType > 2) {
        nf_set_error("nf_fft2_forward: bad arguments");
        return NF_ERR_ARG;
    }
    if ((apodY == nullptr) != (apodX == nullptr)) {
        nf_set_error("nf_fft2_forward: give both apodisation tables or neither");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    if (prec == NF_F32) return p->pf.forward(in, inType, out, batch, apodY, apodX, unitary ? 1 : 0, scale, S(stream));
    return p->pd.forward(in, inType, out, batch, apodY, apodX, unitary ? 1 : 0, scale, S(stream));
}

int nf_fft2_forward_half(nf_plan* p, int prec, const void* in, int inType, void* out, int batch, const double* apodY,
                         const double* apodX, int unitary, void* stream) {
    if (!p || !in || !out || batch < 1 || (inType != NF_IN_REAL64 && inType != NF_IN_REAL32)) {
        nf_set_error("nf_fft2_forward_half: bad arguments (the input must be real)");
        return NF_ERR_ARG;
    }
    if ((apodY == nullptr) != (apodX == nullptr)) {
        nf_set_error("nf_fft2_forward_half: give both apodisation tables or neither");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    if (prec == NF_F32) return p->pf.forward(in, inType, out, batch, apodY, apodX, unitary ? 1 : 0, scale, S(stream), 1);
    return p->pd.forward(in, inType, out, batch, apodY, apodX, unitary ? 1 : 0, scale, S(stream), 1);
}

int nf_fft2_inverse(nf_plan* p, int prec, const void* in, void* work, void* out, int outType, int batch, int unitary,
                    void* stream) {
    if (!p || !in || !work || !out || batch < 1 || outType < 0 || outType > 2) {
        nf_set_error("nf_fft2_inverse: bad arguments");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    if (prec == NF_F32) return p->pf.inverse(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream));
    return p->pd.inverse(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream));
}

int nf_fft2_inverse_fused(nf_plan* p, int prec, const void* in, const float* filt, int nSum, int filtPerBatch,
                          const double* wy, const double* wx, void* work, void* out, int outType, int batch, int unitary,
                          void* stream) {
    if (!p || !in || !work || !out || batch < 1 || outType < 0 || outType > 2 || nSum < 1 || (nSum > 1 && (!filt || batch != 1)) ||
        ((wy == nullptr) != (wx == nullptr))) {
        nf_set_error("nf_fft2_inverse_fused: bad arguments");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    const long long fbs = (filt && filtPerBatch) ? (long long)p->H * p->W : 0;
    if (prec == NF_F32) return p->pf.inverse(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream), filt, nSum, fbs, wy, wx);
    return p->pd.inverse(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream), filt, nSum, fbs, wy, wx);
}

int nf_fft2_inverse_half(nf_plan* p, int prec, const void* in, const double* filtSym, int nSum, int filtPerBatch,
                         const double* wy, const double* wx, const double* planeScale, void* work, void* out, int outType, int batch,
                         int rowLo, int rowHi, int unitary, void* stream) {
    if (!p || !in || !work || !out || batch < 1 || (outType != NF_IN_REAL64 && outType != NF_IN_REAL32) || nSum < 1 ||
        (nSum > 1 && (!filtSym || batch != 1)) || ((wy == nullptr) != (wx == nullptr)) || (planeScale && (nSum > 1 ? nSum : batch) > NF_PLANE_SCALES)) {
        nf_set_error("nf_fft2_inverse_half: bad arguments (the output must be real)");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    const long long fbs = (filtSym && filtPerBatch) ? (long long)p->H * p->W : 0;
    if (prec == NF_F32)
        return p->pf.inverse_half(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream), filtSym, nSum, fbs, wy, wx, nullptr,
                                  nullptr, nullptr, planeScale, rowLo, rowHi);
    return p->pd.inverse_half(in, work, out, outType, batch, unitary ? 1 : 0, scale, S(stream), filtSym, nSum, fbs, wy, wx, nullptr,
                              nullptr, nullptr, planeScale, rowLo, rowHi);
}

int nf_fft2_inverse_half_epi(nf_plan* p, int prec, const void* in, const double* filtSym, int nSum, const double* wy,
                             const double* wx, void* work, void* out, int outType, int unitary, const double* scale_d,
                             const unsigned char* mask_d, const void* zeroRef_d, void* stream) {
    if (!p || !in || !work || !out || (outType != NF_IN_REAL64 && outType != NF_IN_REAL32) || nSum < 1 || (nSum > 1 && !filtSym) ||
        ((wy == nullptr) != (wx == nullptr)) || (mask_d && !scale_d)) {
        nf_set_error("nf_fft2_inverse_half_epi: bad arguments (real output, one map; a mask needs a scale)");
        return NF_ERR_ARG;
    }
    int rc = ensure_prec(p, prec);
    if (rc) return rc;
    double scale = std::sqrt((double)p->H * (double)p->W);
    if (prec == NF_F32)
        return p->pf.inverse_half(in, work, out, outType, 1, unitary ? 1 : 0, scale, S(stream), filtSym, nSum, 0, wy, wx, scale_d,
                                  mask_d, zeroRef_d);
    return p->pd.inverse_half(in, work, out, outType, 1, unitary ? 1 : 0, scale, S(stream), filtSym, nSum, 0, wy, wx, scale_d, mask_d,
                              zeroRef_d);
}

int nf_symmetrise_filter(const float* filt, double* filtSym, int N, int H, int W, void* stream) {
    if (!filt || !filtSym || N < 1 || H < 1 || W < 1) {
        nf_set_error("nf_symmetrise_filter: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_BUILD_FILTER, S(stream), (double)N * H * W * (4.0 + 4.0));
    return nf::symmetrise_filter(filt, filtSym, N, H, W, S(stream));
}

int nf_cross_power(int prec, const void* F, void* NP, int N, long long npix, void* stream) {
    if (!F || !NP || N < 1 || npix < 1) {
        nf_set_error("nf_cross_power: bad arguments");
        return NF_ERR_ARG;
    }
    const double es = prec == NF_F32 ? 4.0 : 8.0;
    nf::ProfScope ps(nf::PT_CROSS_POWER, S(stream), (double)npix * (N * 2 * es + (double)N * N * es));
    if (prec == NF_F32) return nf::cross_power<float>(F, NP, N, npix, S(stream));
    return nf::cross_power<double>(F, NP, N, npix, S(stream));
}

int nf_cross_power_half(int prec, const void* F, void* NP, int N, int H, int W, void* stream) {
    if (!F || !NP || N < 1 || H < 1 || W < 1) {
        nf_set_error("nf_cross_power_half: bad arguments");
        return NF_ERR_ARG;
    }
    const long long npix = (long long)H * W;
    const double es = prec == NF_F32 ? 4.0 : 8.0;
    // each stored mode is read once from HBM (its second use comes out of the cache)
    nf::ProfScope ps(nf::PT_CROSS_POWER, S(stream), (double)npix * (N * es + (double)N * N * es));
    if (prec == NF_F32) return nf::cross_power<float>(F, NP, N, npix, S(stream), H, W);
    return nf::cross_power<double>(F, NP, N, npix, S(stream), H, W);
}

int nf_gaussian_filter(int rtype, const void* in, void* tmp, void* out, int H, int W, int batch, const double* wY, int radY,
                       const double* wX, int radX, void* stream) {
    if (!in || !tmp || !out || !wY || !wX || H < 1 || W < 1 || batch < 1 || radY < 0 || radX < 0) {
        nf_set_error("nf_gaussian_filter: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_GAUSSIAN, S(stream), (double)H * W * batch * 2 * (rtype == NF_F32 ? 4.0 : 8.0));
    return nf::gaussian_filter(rtype, in, tmp, out, H, W, batch, wY, radY, wX, radX, S(stream));
}

int nf_build_filter(int covType, int sigType, const void* cov, const void* Sg, const double* w, float* filt, int N,
                    long long npix, void* stream) {
    if (!cov || !Sg || !w || !filt || N < 1 || N > 6 || npix < 1) {
        nf_set_error("nf_build_filter: bad arguments (1 <= N <= 6)");
        return NF_ERR_ARG;
    }
    nf::WeightVec wv;
    memset(&wv, 0, sizeof(wv));
    for (int i = 0; i < N; ++i) wv.v[i] = w[i];
    nf::ProfScope ps(nf::PT_BUILD_FILTER, S(stream),
                     (double)npix * ((double)N * N * (covType == NF_F32 ? 4.0 : 8.0) + N * (sigType == NF_F32 ? 8.0 : 16.0) + N * 4.0));
    return nf::build_filter(covType, sigType, cov, Sg, wv, filt, N, npix, S(stream));
}

int nf_build_filter_half(int covType, int sigType, const void* cov, const void* Sg, const double* w, float* filt, int N,
                         int H, int W, void* stream) {
    if (!cov || !Sg || !w || !filt || N < 1 || N > 6 || H < 1 || W < 1) {
        nf_set_error("nf_build_filter_half: bad arguments (1 <= N <= 6)");
        return NF_ERR_ARG;
    }
    const long long npix = (long long)H * W;
    nf::WeightVec wv;
    memset(&wv, 0, sizeof(wv));
    for (int i = 0; i < N; ++i) wv.v[i] = w[i];
    nf::ProfScope ps(nf::PT_BUILD_FILTER, S(stream),
                     (double)npix * ((double)N * N * (covType == NF_F32 ? 4.0 : 8.0) + N * (sigType == NF_F32 ? 4.0 : 8.0) + N * 4.0));
    return nf::build_filter(covType, sigType, cov, Sg, wv, filt, N, npix, S(stream), H, W);
}

int nf_build_filter_sym_half(int covType, int sigType, const void* cov, const void* Sg, const double* w, float* filt, double* filtSym,
                             int N, int H, int W, void* stream) {
    if (!cov || !Sg || !w || !filtSym || N < 1 || N > 6 || H < 1 || W < 1) {
        nf_set_error("nf_build_filter_sym_half: bad arguments (1 <= N <= 6)");
        return NF_ERR_ARG;
    }
    const long long npix = (long long)H * W;
    nf::WeightVec wv;
    memset(&wv, 0, sizeof(wv));
    for (int i = 0; i < N; ++i) wv.v[i] = w[i];
    nf::ProfScope ps(nf::PT_BUILD_FILTER, S(stream),
                     (double)npix * ((double)N * N * (covType == NF_F32 ? 4.0 : 8.0) + N * (sigType == NF_F32 ? 4.0 : 8.0) +
                                     (filt ? N * 4.0 : 0.0) + N * 4.0));
    return nf::build_filter_sym(covType, sigType, cov, Sg, wv, filt, filtSym, N, H, W, S(stream));
}

int nf_apply_filter(int prec, const void* F, const float* filt, void* G, int N, int sum, long long npix, void* stream) {
    if (!F || !filt || !G || N < 1 || npix < 1) {
        nf_set_error("nf_apply_filter: bad arguments");
        return NF_ERR_ARG;
    }
    const double cs = prec == NF_F32 ? 8.0 : 16.0;
    nf::ProfScope ps(nf::PT_APPLY_FILTER, S(stream), (double)npix * (N * cs + N * 4.0 + (sum ? 1 : N) * cs));
    if (prec == NF_F32) return nf::apply_filter<float>(F, filt, G, N, sum, npix, S(stream));
    return nf::apply_filter<double>(F, filt, G, N, sum, npix, S(stream));
}

int nf_window_mul(int prec, void* F, const double* wy, const double* wx, int H, int W, int batch, void* stream) {
    if (!F || !wy || !wx || H < 1 || W < 1 || batch < 1) {
        nf_set_error("nf_window_mul: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_WINDOW_MUL, S(stream), (double)H * W * batch * 2 * (prec == NF_F32 ? 8.0 : 16.0));
    if (prec == NF_F32) return nf::window_mul<float>(F, wy, wx, H, W, batch, S(stream));
    return nf::window_mul<double>(F, wy, wx, H, W, batch, S(stream));
}

int nf_paint_radial(float* out, int H, int W, const double* A, const double* B, const double* C, const double* r,
                    const double* prof, const double* slope, int ntab, double r0, double dlnr, double amp, double sign,
                    void* stream) {
    if (!out || !A || !B || !C || !r || !prof || ntab < 2 || H < 1 || W < 1 || !(dlnr > 0) || !(r0 > 0)) {
        nf_set_error("nf_paint_radial: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_PAINT, S(stream), (double)H * W * 4.0);
    return nf::paint_radial(out, H, W, A, B, C, r, prof, slope, ntab, r0, dlnr, amp, sign, S(stream));
}

int nf_paint_radial2(float* out, float* out2, int H, int W, const double* A, const double* B, const double* C, const double* r,
                     const double* prof, const double* slope, const double* prof2, const double* slope2, int ntab, double r0,
                     double dlnr, double amp, double sign, double amp2, double sign2, void* stream) {
    if (!out || !out2 || !A || !B || !C || !r || !prof || !prof2 || ntab < 2 || H < 1 || W < 1 || !(dlnr > 0) || !(r0 > 0)) {
        nf_set_error("nf_paint_radial2: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_PAINT, S(stream), (double)H * W * 8.0);
    return nf::paint_radial2(out, out2, H, W, A, B, C, r, prof, slope, prof2, slope2, ntab, r0, dlnr, amp, sign, amp2, sign2,
                             S(stream));
}

int nf_scale_mask(int vtype, const float* in, void* out, double norm, const unsigned char* psMask, long long n, void* stream) {
    if (!in || !out || n < 1) {
        nf_set_error("nf_scale_mask: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_SCALE_MASK, S(stream), (double)n * (4.0 + (vtype == NF_F32 ? 4.0 : 8.0) + (psMask ? 1.0 : 0.0)));
    return nf::scale_mask(vtype, in, out, norm, nullptr, psMask, n, S(stream));
}

int nf_scale_mask_dev(int vtype, const float* in, void* out, const double* norm_d, const unsigned char* psMask, long long n,
                      void* stream) {
    if (!in || !out || !norm_d || n < 1) {
        nf_set_error("nf_scale_mask_dev: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_SCALE_MASK, S(stream), (double)n * (4.0 + (vtype == NF_F32 ? 4.0 : 8.0) + (psMask ? 1.0 : 0.0)));
    return nf::scale_mask(vtype, in, out, 1.0, norm_d, psMask, n, S(stream));
}

size_t nf_noise_workspace_bytes(int H, int W, int ny, int nx, int numBins) {
    nf::NoiseGrid g;
    memset(&g, 0, sizeof(g));
    g.ny = ny;
    g.nx = nx;
    g.numBins = numBins;
    return nf::noise_workspace_bytes(g, H, W);
}

int nf_noise_map(int vtype, const void* filtered, const void* weights, int wtype, float* rms, int H, int W, int ny, int nx,
                 const int* yEdges, const int* xEdges, int overlap, int numBins, const double* quantiles, int estimator,
                 void* work, size_t workBytes, int tablesValid, void* stream) {
    if (!filtered || !weights || !rms || !yEdges || !xEdges || !work || ny < 1 || nx < 1 || overlap < 0 || numBins < 1 ||
        numBins > 16 || (numBins > 1 && !quantiles)) {
        nf_set_error("nf_noise_map: bad arguments");
        return NF_ERR_ARG;
    }
    nf::NoiseGrid g;
    memset(&g, 0, sizeof(g));
    g.ny = ny;
    g.nx = nx;
    g.overlap = overlap;
    g.numBins = numBins;
    g.estimator = estimator;
    g.pctQ = 68.3 / 100.0;      // np.percentile(a, 68.3): q = np.true_divide(68.3, 100)
    g.yEdges = yEdges;
    g.xEdges = xEdges;
    for (int i = 0; i < numBins && quantiles; ++i) g.quantiles[i] = quantiles[i];
    // algorithmic bytes: the map in, the noise map out, and the weights unless the kept tables stand for them
    nf::ProfScope ps(nf::PT_NOISE_MAP, S(stream),
                     (double)H * W * ((vtype == NF_F32 ? 4.0 : 8.0) + (tablesValid ? 0.0 : (wtype == NF_F32 ? 4.0 : 8.0)) + 4.0));
    return nf::noise_map(vtype, filtered, weights, wtype, rms, H, W, g, work, workBytes, tablesValid != 0, S(stream));
}

int nf_sn_cached(int vtype, const void* data, const float* rms, double* sn, long long n, void* stream) {
    if (!data || !rms || !sn || n < 1) {
        nf_set_error("nf_sn_cached: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_FINALIZE, S(stream), (double)n * ((vtype == NF_F32 ? 4.0 : 8.0) + 4.0 + 8.0));
    return nf::sn_cached(vtype, data, rms, sn, n, S(stream));
}

size_t nf_noise_smart_workspace_bytes(void) { return nf::noise_smart_workspace_bytes(); }

int nf_noise_map_smart(int vtype, const void* filtered, const void* weights, int wtype, float* rms, long long n, int numBins,
                       void* work, size_t workBytes, void* stream) {
    if (!filtered || !weights || !rms || !work || n < 1) {
        nf_set_error("nf_noise_map_smart: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_NOISE_MAP, S(stream), (double)n * ((vtype == NF_F32 ? 4.0 : 8.0) + (wtype == NF_F32 ? 4.0 : 8.0) + 4.0));
    return nf::noise_map_smart(vtype, filtered, weights, wtype, rms, n, numBins, work, workBytes, S(stream));
}

int nf_sum_planes(const float* in, float* out, int N, long long n, void* stream) {
    if (!in || !out || N < 1 || n < 1) {
        nf_set_error("nf_sum_planes: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_APPLY_FILTER, S(stream), (double)n * (N + 1) * 4.0);
    return nf::sum_planes(in, out, N, n, S(stream));
}

int nf_median_stack(int rtype, const void* in, void* out, int N, long long n, void* stream) {
    if (!in || !out || N < 1 || N > 8 || n < 1) {
        nf_set_error("nf_median_stack: bad arguments (1 <= N <= 8)");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_MEDIAN_STACK, S(stream), (double)n * (N + 1) * (rtype == NF_F32 ? 4.0 : 8.0));
    return nf::median_stack(rtype, in, out, N, n, S(stream));
}

int nf_edge_check(int vtype, const void* filtered, const unsigned char* psMask, unsigned char* tmp, unsigned char* out, int H,
                  int W, int size, void* stream) {
    if (!filtered || !tmp || !out || H < 1 || W < 1 || size < 1 || size > H || size > W) {
        nf_set_error("nf_edge_check: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_EDGE_CHECK, S(stream), (double)H * W * ((vtype == NF_F32 ? 4.0 : 8.0) + 2.0));
    return nf::edge_check(vtype, filtered, psMask, tmp, out, H, W, size, S(stream));
}

int nf_finalize(int vtype, void* filtered, const float* rms, const unsigned char* surveyMask, const unsigned char* psMask,
                const unsigned char* edgeCheck, void* sn, float* rmsOut, unsigned char* maskOut, int H, int W, int apodPix,
                int mulSurvey, unsigned long long* snChunkRange, int* zeroRows, void* stream) {
    if (!filtered || !rms || !sn || !rmsOut || !maskOut || H < 1 || W < 1) {
        nf_set_error("nf_finalize: bad arguments");
        return NF_ERR_ARG;
    }
    // map + noise map + three masks in; S/N map + mask out (the filtered map and the noise map are stored only where they change)
    nf::ProfScope ps(nf::PT_FINALIZE, S(stream), (double)H * W * (2 * (vtype == NF_F32 ? 4.0 : 8.0) + 4.0 + 3.0 + 1.0));
    return nf::finalize(vtype, filtered, rms, surveyMask, psMask, edgeCheck, sn, rmsOut, maskOut, H, W, apodPix, mulSurvey,
                        snChunkRange, zeroRows, S(stream));
}

int nf_finalize_cached(int vtype, void* filtered, const float* cachedRms, const unsigned char* surveyMask, const unsigned char* psMask,
                       const unsigned char* edgeCheck, double* sn, unsigned char* maskOut, int H, int W, int apodPix, int mulSurvey,
                       unsigned long long* snChunkRange, int* zeroRows, void* stream) {
    if (!filtered || !cachedRms || !sn || !maskOut || H < 1 || W < 1 || (vtype != NF_F32 && vtype != NF_F64)) {
        nf_set_error("nf_finalize_cached: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_FINALIZE, S(stream), (double)H * W * ((vtype == NF_F32 ? 4.0 : 8.0) + 4.0 + 3.0 + 8.0 + 1.0));
    return nf::finalize_cached(vtype, filtered, cachedRms, surveyMask, psMask, edgeCheck, sn, maskOut, H, W, apodPix, mulSurvey,
                               snChunkRange, zeroRows, S(stream));
}

int nf_load_tile(int rtype, const void* map, int wtype, const void* weights, const unsigned char* surveyMask, const unsigned char* psMask,
                 const unsigned char* flagMask, int srcH, int srcW, int x0, int y0, int H, int W, int ax0, int ax1, int ay0, int ay1,
                 void* dataOut, void* weightsOut, unsigned char* surveyOut, unsigned char* psOut, unsigned char* flagOut, void* stream) {
    if (!map || !dataOut || !weightsOut || !surveyOut || !psOut || !flagOut || H < 1 || W < 1 || x0 < 0 || y0 < 0 ||
        (long long)x0 + W > srcW || (long long)y0 + H > srcH || (rtype != NF_F32 && rtype != NF_F64) || (wtype != NF_F32 && wtype != NF_F64)) {
        nf_set_error("nf_load_tile: bad arguments (the clip must lie inside the source map)");
        return NF_ERR_ARG;
    }
    if (H > 65535) {
        nf_set_error("nf_load_tile: more than 65535 rows in one tile (one workgroup row per tile row)");
        return NF_ERR_SIZE;
    }
    return nf::load_tile(rtype, map, wtype, weights, surveyMask, psMask, flagMask, srcW, x0, y0, H, W, ax0, ax1, ay0, ay1, dataOut, weightsOut,
                         surveyOut, psOut, flagOut, S(stream));
}

int nf_inject_model(int baseType, const void* base, int modelType, const void* model, int wType, const void* weights, void* out,
                    long long n, void* stream) {
    if (!base || !model || !out || n < 1 || (baseType != NF_F32 && baseType != NF_F64) || (modelType != NF_F32 && modelType != NF_F64) ||
        (weights && wType != NF_F32 && wType != NF_F64)) {
        nf_set_error("nf_inject_model: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::inject_model(baseType, base, modelType, model, wType, weights, out, n, S(stream));
}

int nf_sn_chunks(int H, int W) { return (H < 1 || W < 1) ? 0 : nf_div_up((long long)H * W, NF_SN_CHUNK); }

int nf_restore_zeros(int vtype, void* data, const void* ref, long long n, void* stream) {
    if (!data || !ref || n < 1) {
        nf_set_error("nf_restore_zeros: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_RESTORE_ZEROS, S(stream), (double)n * 3 * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::restore_zeros(vtype, data, ref, n, S(stream));
}

int nf_subtract(int rtype, const void* a, const void* b, void* out, long long n, void* stream) {
    if (!a || !b || !out || n < 1) {
        nf_set_error("nf_subtract: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_SUBTRACT, S(stream), (double)n * 3 * (rtype == NF_F32 ? 4.0 : 8.0));
    return nf::subtract(rtype, a, b, out, n, S(stream));
}

int nf_convolve2d(const float* in, float* out, int H, int W, const double* kernFlipped, int kh, int kw, void* stream) {
    if (!in || !out || !kernFlipped || H < 1 || W < 1 || kh < 1 || kw < 1 || in == out) {
        nf_set_error("nf_convolve2d: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_CONVOLVE2D, S(stream), (double)H * W * 8.0);
    return nf::convolve2d(NF_F32, in, out, H, W, kernFlipped, kh, kw, S(stream));
}

int nf_convolve2d_f64(const double* in, double* out, int H, int W, const double* kernFlipped, int kh, int kw, void* stream) {
    if (!in || !out || !kernFlipped || H < 1 || W < 1 || kh < 1 || kw < 1 || in == out) {
        nf_set_error("nf_convolve2d_f64: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_CONVOLVE2D, S(stream), (double)H * W * 16.0);
    return nf::convolve2d(NF_F64, in, out, H, W, kernFlipped, kh, kw, S(stream));
}

int nf_median_filter(int rtype, const void* in, void* out, int H, int W, int size, void* stream) {
    if (!in || !out || in == out || H < 1 || W < 1 || (rtype != NF_F32 && rtype != NF_F64)) {
        nf_set_error("nf_median_filter: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::median_filter(rtype, in, out, H, W, size, S(stream));
}

int nf_disc_fill(double* rmap, int H, int W, const double* rec, const int* box, const int* boxHost, void* stream) {
    if (!rmap || !rec || !box || !boxHost || boxHost[0] < 0 || boxHost[1] > W || boxHost[2] < 0 || boxHost[3] > H) {
        nf_set_error("nf_disc_fill: bad arguments (box outside the map)");
        return NF_ERR_ARG;
    }
    const int n = (boxHost[1] - boxHost[0]) * (boxHost[3] - boxHost[2]);
    return nf::disc_fill(rmap, W, rec, box, n > 0 ? n : 0, S(stream));
}

int nf_disc_apply(int rtype, void* data, const void* bck, unsigned char* surveyMask, unsigned char* psMask, int H, int W,
                  const double* recs, const int* boxes, const int* boxesHost, int nBoxes, double limit, int times60,
                  const double* rmap, void* stream) {
    if (!data || !bck || !surveyMask || !psMask || !recs || !boxes || !boxesHost || nBoxes < 0 || (rtype != NF_F32 && rtype != NF_F64)) {
        nf_set_error("nf_disc_apply: bad arguments");
        return NF_ERR_ARG;
    }
    for (int k = 0; k < nBoxes; ++k) {
        const int* b = boxesHost + 4 * k;
        if (b[0] < 0 || b[1] > W || b[2] < 0 || b[3] > H) {
            nf_set_error("nf_disc_apply: box outside the map");
            return NF_ERR_ARG;
        }
    }
    return nf::disc_apply(rtype, data, bck, surveyMask, psMask, W, recs, boxes, nBoxes, limit, times60, rmap, S(stream));
}

int nf_paint_catalog(float* out, int H, int W, int nObj, const double* obj, const int* win, const int* winHost, const double* geom,
                     long long geomLen, const double* prof, int nProf, int ntab, int accumulate, void* stream) {
    if (!out || !obj || !win || !winHost || !geom || !prof || H < 1 || W < 1 || nObj < 1 || ntab < 2 || nProf < 1) {
        nf_set_error("nf_paint_catalog: bad arguments");
        return NF_ERR_ARG;
    }
    for (int k = 0; k < nObj; ++k) {
        const int* w = winHost + 8 * k;
        const long long rows = w[3] - w[2], cols = w[1] - w[0];
        if (w[0] < 0 || w[1] > W || w[2] < 0 || w[3] > H || rows < 0 || cols < 0 || w[4] < 0 || w[5] < 0 || w[4] + 2 * rows > geomLen ||
            w[5] + cols > geomLen || w[6] < 0 || w[6] >= nProf) {
            nf_set_error("nf_paint_catalog: window or table offsets outside their arrays");
            return NF_ERR_ARG;
        }
    }
    return nf::paint_catalog(out, H, W, nObj, obj, win, geom, prof, ntab, accumulate, S(stream));
}

int nf_resample_bilinear(const float* in, float* out, int N, int H, int W, int H2, int W2, const double* yIdx, const double* xIdx,
                         void* stream) {
    if (!in || !out || !yIdx || !xIdx || N < 1 || H < 1 || W < 1 || H2 < 1 || W2 < 1 || H2 > 65535 || N > 65535) {
        nf_set_error("nf_resample_bilinear: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::resample_bilinear(in, out, N, H, W, H2, W2, yIdx, xIdx, S(stream));
}

int nf_add_model(int rtype, void* data, const float* model, unsigned char* flagMask, long long n, int subtract, void* stream) {
    if (!data || !model || n < 1 || (rtype != NF_F32 && rtype != NF_F64)) {
        nf_set_error("nf_add_model: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::add_model(rtype, data, model, flagMask, n, subtract, S(stream));
}

int nf_label_components(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work,
                        int* numObjects, void* stream) {
    if (!sn || !labels || !work || !numObjects || H < 1 || W < 1) {
        nf_set_error("nf_label_components: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_LABEL, S(stream), (double)H * W * ((vtype == NF_F32 ? 4.0 : 8.0) + 4.0));
    return nf::label_components(vtype, sn, threshold, negate, labels, H, W, work, numObjects, nullptr, S(stream));
}

int nf_object_stats(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* stats,
                    void* stream) {
    if (!sn || !labels || !stats || H < 1 || W < 1 || numObjects < 0) {
        nf_set_error("nf_object_stats: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_OBJECT_STATS, S(stream), (double)H * W * ((vtype == NF_F32 ? 4.0 : 8.0) + 4.0));
    return nf::object_stats(vtype, sn, negate, labels, H, W, numObjects, stats, nullptr, S(stream));
}

long long nf_label_chunk_flags_offset(int H, int W) { return nf::label_chunk_flags_offset(H, W); }

int nf_label_components_kept(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work,
                             int* dirty, int* numObjects, void* stream) {
    if (!sn || !labels || !work || !dirty || !numObjects || H < 1 || W < 1) {
        nf_set_error("nf_label_components_kept: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_LABEL, S(stream), (double)H * W * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::label_components(vtype, sn, threshold, negate, labels, H, W, work, numObjects, dirty, S(stream));
}

int nf_object_stats_chunks(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* stats,
                           const int* chunkFlags, void* stream) {
    if (!sn || !labels || !stats || !chunkFlags || H < 1 || W < 1 || numObjects < 0) {
        nf_set_error("nf_object_stats_chunks: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_OBJECT_STATS, S(stream), (double)H * W * 4.0 / 1024.0);
    return nf::object_stats(vtype, sn, negate, labels, H, W, numObjects, stats, chunkFlags, S(stream));
}

int nf_label_and_stats(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work, int* dirty,
                       int maxObjects, double* stats, const unsigned long long* snChunkRange, void* stream) {
    if (!sn || !labels || !work || !stats || H < 1 || W < 1 || maxObjects < 0 || (vtype != NF_F32 && vtype != NF_F64)) {
        nf_set_error("nf_label_and_stats: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_LABEL, S(stream), (double)H * W * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::label_and_stats(vtype, sn, threshold, negate, labels, H, W, work, dirty, maxObjects, stats, snChunkRange, S(stream));
}

int nf_label_and_stats_batch(int vtype, int numMaps, const void* const* sn, double threshold, int negate, int* const* labels, int H,
                             int W, int* const* work, int* const* dirty, int maxObjects, double* const* stats,
                             const unsigned long long* const* snChunkRange, void* stream) {
    if (!sn || !labels || !work || !stats || numMaps < 1 || H < 1 || W < 1 || maxObjects < 0 || (vtype != NF_F32 && vtype != NF_F64)) {
        nf_set_error("nf_label_and_stats_batch: bad arguments");
        return NF_ERR_ARG;
    }
    for (int m = 0; m < numMaps; ++m)
        if (!sn[m] || !labels[m] || !work[m] || !stats[m]) {
            nf_set_error("nf_label_and_stats_batch: a map, label, work or record pointer is NULL");
            return NF_ERR_ARG;
        }
    nf::ProfScope ps(nf::PT_LABEL, S(stream), (double)numMaps * H * W * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::label_and_stats_batch(vtype, numMaps, sn, threshold, negate, labels, H, W, work, dirty, maxObjects, stats, snChunkRange,
                                     S(stream));
}

int nf_object_shapes(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* shapes,
                     void* stream) {
    if (!sn || !labels || !shapes || H < 1 || W < 1 || numObjects < 0 || (vtype != NF_F32 && vtype != NF_F64)) {
        nf_set_error("nf_object_shapes: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::object_shapes(vtype, sn, negate, labels, H, W, numObjects, shapes, S(stream));
}

int nf_rms_table(const float* rms, const unsigned char* areaMask, const int* flagMask, int maxFlags, const unsigned char* intersect,
                 const unsigned long long* rowAreaFix, int H, int W, unsigned* keys, unsigned long long* counts,
                 unsigned long long* areas, int tableSize, int* overflow, void* stream) {
    if (!rms || !rowAreaFix || !keys || !counts || !areas || !overflow || H < 1 || W < 1) {
        nf_set_error("nf_rms_table: bad arguments");
        return NF_ERR_ARG;
    }
    return nf::rms_table(rms, areaMask, flagMask, maxFlags, intersect, rowAreaFix, H, W, keys, counts, areas, tableSize, overflow, S(stream));
}

size_t nf_xmatch_workspace_bytes(int n) { return n > 0 ? nf::xmatch_workspace_bytes(n) : 0; }

int nf_xmatch_self(const double* unit_d, const double* snr_d, int n, double radiusDeg, int* count_d, int* best_d,
                   unsigned char* drop_d, int* nearFlag_d, void* work_d, size_t workBytes, void* stream) {
    if (!unit_d || !snr_d || !count_d || !best_d || !drop_d || !nearFlag_d || !work_d || n < 1 || !(radiusDeg > 0.0)) {
        nf_set_error("nf_xmatch_self: bad arguments");
        return NF_ERR_ARG;
    }
    const double lim = 2.0 * std::sin(radiusDeg * 3.14159265358979323846 / 180.0 / 2.0);
    const double lo = lim * (1.0 - 1e-9), hi = lim * (1.0 + 1e-9);
    return nf::xmatch_self(unit_d, snr_d, n, lo * lo, hi * hi, count_d, best_d, drop_d, nearFlag_d, work_d, workBytes, S(stream));
}

int nf_xmatch_static_max_rows(void) { return NF_XM_STATIC_MAX_ROWS; }

int nf_xmatch_self_static(const double* unit_d, const double* snr_d, int n, double radiusDeg, int* count_d, int* best_d,
                          unsigned char* drop_d, int* nearFlag_d, void* stream) {
    if (!unit_d || !snr_d || !count_d || !best_d || !drop_d || !nearFlag_d || n < 1 || !(radiusDeg > 0.0)) {
        nf_set_error("nf_xmatch_self_static: bad arguments");
        return NF_ERR_ARG;
    }
    const double lim = 2.0 * std::sin(radiusDeg * 3.14159265358979323846 / 180.0 / 2.0);
    const double lo = lim * (1.0 - 1e-9), hi = lim * (1.0 + 1e-9);
    return nf::xmatch_self_static(unit_d, snr_d, n, lo * lo, hi * hi, count_d, best_d, drop_d, nearFlag_d, S(stream));
}

int nf_map_max(int vtype, const void* map, long long n, double* out, void* stream) {
    if (!map || !out || n < 1) {
        nf_set_error("nf_map_max: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_MAP_MAX, S(stream), (double)n * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::map_max(vtype, map, n, out, S(stream));
}

int nf_ring_flags(const int* ringLabels, const double* ringStats, int numRing, int minRingPix, int H, int W,
                  unsigned char* isRing, void* stream) {
    if (!ringLabels || !ringStats || !isRing || numRing < 0) {
        nf_set_error("nf_ring_flags: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_RING_FLAGS, S(stream), (double)numRing * 49.0);
    return nf::ring_flags(ringLabels, ringStats, numRing, minRingPix, H, W, isRing, S(stream));
}

int nf_spline_eval(int vtype, const void* map, int negate, int H, int W, const double* yx, int nPts, int halfWin,
                   int pointsClass, const int* edgeSel, int nEdge, double* out, void* stream) {
    if (!map || (nPts > 0 && (!yx || !out)) || nPts < 0 || halfWin < 2 || pointsClass < 0 || pointsClass > 2 ||
        (edgeSel && (nEdge < 1 || nEdge > nPts || pointsClass != 0))) {
        nf_set_error("nf_spline_eval: bad arguments");
        return NF_ERR_ARG;
    }
    nf::ProfScope ps(nf::PT_SPLINE, S(stream), (double)nPts * (2.0 * halfWin + 1) * (2.0 * halfWin + 1) * (vtype == NF_F32 ? 4.0 : 8.0));
    return nf::spline_eval(vtype, map, negate, H, W, yx, nPts, halfWin, pointsClass, edgeSel, nEdge, out, S(stream));
}

int nf_spline_eval_batch(int vtype, int nJobs, const void* const* maps, const int* negate, const double* const* yx, const int* nPts,
                         const int* const* edgeSel, const int* nEdge, double* const* out, int H, int W, int halfWin, void* stream) {
    if (nJobs < 0 || nJobs > NF_SPL_MAXJOBS || !maps || !negate || !yx || !nPts || !edgeSel || !nEdge || !out || H < 4 || W < 4 ||
        (vtype != NF_F32 && vtype != NF_F64)) {
        nf_set_error("nf_spline_eval_batch: bad arguments (at most 32 jobs)");
        return NF_ERR_ARG;
    }
    nf::SplineBatch b;
    for (int j = 0; j < NF_SPL_MAXJOBS; ++j) {
        const bool live = j < nJobs;
        if (live && nPts[j] > 0 && (!maps[j] || !yx[j] || !out[j] || nEdge[j] < 0 || nEdge[j] > nPts[j])) {
            nf_set_error("nf_spline_eval_batch: bad job");
            return NF_ERR_ARG;
        }
        b.map[j] = live ? maps[j] : nullptr;
        b.yx[j] = live ? yx[j] : nullptr;
        b.edge[j] = live ? edgeSel[j] : nullptr;
        b.out[j] = live ? out[j] : nullptr;
        b.n[j] = live ? nPts[j] : 0;
        b.nEdge[j] = live ? (edgeSel[j] ? nEdge[j] : nPts[j]) : 0;
        b.negate[j] = live ? negate[j] : 0;
    }
    double bytes = 0.0;
    for (int j = 0; j < nJobs; ++j) bytes += (double)nPts[j] * (2 * halfWin + 4) * (2 * halfWin + 4) * (vtype == NF_F32 ? 4.0 : 8.0);
    nf::ProfScope ps(nf::PT_SPLINE, S(stream), bytes);
    return nf::spline_eval_batch(vtype, b, nJobs, H, W, halfWin, S(stream));
}

}  // extern "C"

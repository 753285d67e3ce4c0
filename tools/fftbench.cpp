// Micro-benchmark of the 2-D FFT entry points through the C ABI (include/nemofilt.h): per-pass device time from the
// library's own HIP-event profiler, algorithmic GB/s (each pass reads and writes the plane once).
//   fftbench H W prec(0=f32,1=f64) batch reps mode(0 = complex, 1 = real map <-> full spectrum, 2 = real map <-> half-stored spectrum,
//                                                       3 = the matched filter's data inverse: half-stored spectrum x symmetrised filter x window)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/nemofilt.h"

#define CK(x) do { int _r = (x); if (_r != 0) { fprintf(stderr, "%s failed: %d %s\n", #x, _r, nf_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

int main(int argc, char** argv) {
    int H = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 4096;
    int prec = argc > 3 ? atoi(argv[3]) : 1, batch = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 10;
    int mode = argc > 6 ? atoi(argv[6]) : 0;
    size_t es = prec ? 16 : 8;
    size_t n = (size_t)H * W * batch;
    void *a, *b, *c;
    HK(hipMalloc(&a, n * es));
    HK(hipMalloc(&b, n * es));
    HK(hipMalloc(&c, n * es));
    std::vector<float> h(n * es / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.0f;
    if (prec) { double* d = (double*)h.data(); for (size_t i = 0; i < n * 2; ++i) d[i] = (double)rand() / RAND_MAX - 0.5; }
    else for (size_t i = 0; i < n * 2; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    HK(hipMemcpy(a, h.data(), n * es, hipMemcpyHostToDevice));
    nf_plan* plan;
    CK(nf_plan_create(H, W, &plan));
    if (mode != 0) {
        // real maps of the precision's own real type: a [batch][H][W] real array at the start of `a`, real output into `c`
        const int rt = prec ? NF_IN_REAL64 : NF_IN_REAL32;
        nf_profile_enable(1);
        double *filtSym = nullptr, *wy = nullptr, *wx = nullptr;
        void* work = nullptr;
        for (int i = 0; i < reps + 2; ++i) {
            if (i == 2) { HK(hipDeviceSynchronize()); nf_profile_enable(0); nf_profile_enable(1); }
            if (mode == 1) {
                CK(nf_fft2_forward(plan, prec, a, rt, b, batch, nullptr, nullptr, 1, nullptr));
                CK(nf_fft2_inverse(plan, prec, b, b, c, rt, batch, 1, nullptr));
            } else if (mode == 2) {
                CK(nf_fft2_forward_half(plan, prec, a, rt, b, batch, nullptr, nullptr, 1, nullptr));
                CK(nf_fft2_inverse_half(plan, prec, b, nullptr, 1, 0, nullptr, nullptr, nullptr, b, c, rt, batch, 0, 0, 1, nullptr));
            } else {
                // mode 3: the matched filter's data inverse - symmetrised filter (all ones), window tables (all ones), a plane factor,
                // the spectrum kept (work buffer apart from the input)
                if (!filtSym) {
                    std::vector<double> ones((size_t)H * W, 1.0);
                    HK(hipMalloc(&filtSym, ones.size() * 8)); HK(hipMemcpy(filtSym, ones.data(), ones.size() * 8, hipMemcpyHostToDevice));
                    HK(hipMalloc(&wy, H * 8)); HK(hipMemcpy(wy, ones.data(), H * 8, hipMemcpyHostToDevice));
                    HK(hipMalloc(&wx, W * 8)); HK(hipMemcpy(wx, ones.data(), W * 8, hipMemcpyHostToDevice));
                    HK(hipMalloc(&work, n * es));
                    CK(nf_fft2_forward_half(plan, prec, a, rt, b, batch, nullptr, nullptr, 1, nullptr));
                }
                const double psc[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
                CK(nf_fft2_inverse_half(plan, prec, b, filtSym, 1, 0, wy, wx, psc, work, c, rt, batch, 0, 0, 1, nullptr));
            }
        }
        HK(hipDeviceSynchronize());
        std::vector<float> r(n * es / 8);
        HK(hipMemcpy(r.data(), c, n * es / 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        if (prec) { double *x = (double*)h.data(), *y = (double*)r.data(); for (size_t i = 0; i < n; ++i) { double e = x[i] - y[i]; if (e < 0) e = -e; if (e > maxerr) maxerr = e; } }
        else for (size_t i = 0; i < n; ++i) { double e = h[i] - r[i]; if (e < 0) e = -e; if (e > maxerr) maxerr = e; }
        printf("H=%d W=%d prec=%s batch=%d mode=%s roundtrip_maxerr=%.3g\n", H, W, prec ? "f64" : "f32", batch, mode == 1 ? "real/full" : (mode == 2 ? "real/half" : "filtered inverse of a half-stored spectrum"), maxerr);
        for (int t = 0; t < nf_profile_num_tags(); ++t) {
            long long cnt; double ms, by;
            CK(nf_profile_read(t, &cnt, &ms, &by));
            if (cnt > 0) printf("  %-14s calls=%lld avg_us=%.1f GB/s=%.0f\n", nf_profile_tag_name(t), cnt, 1e3 * ms / cnt, by / 1e9 / (ms / 1e3));
        }
        nf_plan_destroy(plan);
        return 0;
    }
    for (int i = 0; i < 2; ++i) {
        CK(nf_fft2_forward(plan, prec, a, NF_IN_COMPLEX, b, batch, nullptr, nullptr, 1, nullptr));
        CK(nf_fft2_inverse(plan, prec, b, c, c, NF_IN_COMPLEX, batch, 1, nullptr));
    }
    HK(hipDeviceSynchronize());
    // round trip error
    std::vector<float> r(n * es / 4);
    HK(hipMemcpy(r.data(), c, n * es, hipMemcpyDeviceToHost));
    double maxerr = 0;
    if (prec) { double *x = (double*)h.data(), *y = (double*)r.data(); for (size_t i = 0; i < n * 2; ++i) { double e = x[i] - y[i]; if (e < 0) e = -e; if (e > maxerr) maxerr = e; } }
    else for (size_t i = 0; i < n * 2; ++i) { double e = h[i] - r[i]; if (e < 0) e = -e; if (e > maxerr) maxerr = e; }
    nf_profile_enable(1);
    for (int i = 0; i < reps; ++i) {
        CK(nf_fft2_forward(plan, prec, a, NF_IN_COMPLEX, b, batch, nullptr, nullptr, 1, nullptr));
        CK(nf_fft2_inverse(plan, prec, b, c, c, NF_IN_COMPLEX, batch, 1, nullptr));
    }
    HK(hipDeviceSynchronize());
    printf("H=%d W=%d prec=%s batch=%d roundtrip_maxerr=%.3g\n", H, W, prec ? "f64" : "f32", batch, maxerr);
    for (int t = 0; t < nf_profile_num_tags(); ++t) {
        long long cnt; double ms, by;
        CK(nf_profile_read(t, &cnt, &ms, &by));
        if (cnt > 0) printf("  %-14s calls=%lld avg_us=%.1f GB/s=%.0f\n", nf_profile_tag_name(t), cnt, 1e3 * ms / cnt, by / 1e9 / (ms / 1e3));
    }
    nf_profile_enable(0);
    nf_plan_destroy(plan);
    return 0;
}

// Experiment ("not tried" until round 3): the column transform of a 4096-row half-stored spectrum as TWO
// passes of 64-point transforms (4096 = 64 x 64) over tiles of CT adjacent columns (CT x 16 B = 256 / 512-byte runs), against
// the library's single-pass column kernel whose 64 KB columns allow 32-byte runs only.
//   n = 64 a + r1, k = k1 + 64 k2:  Y[r1][k1] = w4096^(r1 k1) * sum_a x[64 a + r1] w64^(a k1)      (pass 1, twiddle at the store)
//                                   X[k1 + 64 k2] = sum_r1 Y[r1][k1] w64^(r1 k2)                    (pass 2)
// A workgroup owns (column tile, r1) in pass 1 and (column tile, k1) in pass 2: 64 rows x CT columns, 8 x CT threads, a thread
// does a radix-8 butterfly, one LDS exchange, a second radix-8 butterfly.
//   col2pass [prec 0=f32 1=f64] [reps] [ncols]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); exit(1); } } while (0)

template <typename R> struct C2 { R x, y; };
template <typename R> __device__ __forceinline__ C2<R> operator+(C2<R> a, C2<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> __device__ __forceinline__ C2<R> operator-(C2<R> a, C2<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> __device__ __forceinline__ C2<R> cmul(C2<R> a, C2<R> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename R> __device__ __forceinline__ C2<R> mulmi(C2<R> a) { return {a.y, -a.x}; }   // a * (-i)

template <typename R> __device__ __forceinline__ void dft4(C2<R>& x0, C2<R>& x1, C2<R>& x2, C2<R>& x3) {
    C2<R> t0 = x0 + x2, t1 = x0 - x2, t2 = x1 + x3, t3 = mulmi(x1 - x3);
    x0 = t0 + t2; x1 = t1 + t3; x2 = t0 - t2; x3 = t1 - t3;
}
template <typename R> __device__ __forceinline__ void dft8(C2<R>* v) {
    const R h = (R)0.70710678118654752440;
    C2<R> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3); dft4(o0, o1, o2, o3);
    C2<R> w1 = {h * (o1.x + o1.y), h * (o1.y - o1.x)};          // o1 * (1 - i)/sqrt2
    C2<R> w2 = mulmi(o2);
    C2<R> w3 = {h * (o3.y - o3.x), h * (-o3.x - o3.y)};         // o3 * (-1 - i)/sqrt2
    v[0] = e0 + o0; v[4] = e0 - o0; v[1] = e1 + w1; v[5] = e1 - w1;
    v[2] = e2 + w2; v[6] = e2 - w2; v[3] = e3 + w3; v[7] = e3 - w3;
}

// SIN / SOUT: rows of the tile lie 64 apart (blk + 64 i) or one after the other (64 blk + i).
template <typename R, int CT, bool SIN, bool SOUT, bool TWID, bool FILT>
__global__ __launch_bounds__(CT * 8) void k_pass(const C2<R>* __restrict__ src, C2<R>* __restrict__ dst, const double* __restrict__ filt,
                                                 const C2<R>* __restrict__ tw, int pitch, int ncols, int tile0) {
    __shared__ C2<R> lds[64 * CT];
    const int tile = blockIdx.x + tile0, blk = blockIdx.y;
    const int c = threadIdx.x % CT, j = threadIdx.x / CT;
    const int col = tile * CT + c;
    const bool ok = col < ncols;
    C2<R> v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int a = j + 8 * m;
        const size_t row = SIN ? (size_t)(blk + 64 * a) : (size_t)(64 * blk + a);
        v[m] = ok ? src[row * pitch + col] : C2<R>{0, 0};
        if (FILT) { const R f = ok ? (R)filt[row * pitch + col] : (R)0; v[m].x *= f; v[m].y *= f; }
    }
    dft8(v);
#pragma unroll
    for (int p = 1; p < 8; ++p) v[p] = cmul(v[p], tw[(64 * j * p) & 4095]);
#pragma unroll
    for (int p = 0; p < 8; ++p) lds[(j * 8 + p) * CT + c] = v[p];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = lds[(jj * 8 + j) * CT + c];
    dft8(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = j + 8 * q;
        if (TWID) v[q] = cmul(v[q], tw[(blk * k) & 4095]);
        const size_t row = SOUT ? (size_t)(blk + 64 * k) : (size_t)(64 * blk + k);
        if (ok) dst[row * pitch + col] = v[q];
    }
}

template <typename R> __global__ void k_copy(const C2<R>* __restrict__ src, C2<R>* __restrict__ dst, int pitch, int ncols) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < ncols) dst[(size_t)blockIdx.y * pitch + col] = src[(size_t)blockIdx.y * pitch + col];
}

template <typename R, int CT> int run(int reps, int ncols) {
    const int H = 4096, pitch = 4096;
    const size_t n = (size_t)H * pitch;
    C2<R>*a, *b, *c, *tw; double* filt;
    HK(hipMalloc(&a, n * sizeof(C2<R>))); HK(hipMalloc(&b, n * sizeof(C2<R>))); HK(hipMalloc(&c, n * sizeof(C2<R>)));
    HK(hipMalloc(&tw, 4096 * sizeof(C2<R>))); HK(hipMalloc(&filt, n * sizeof(double)));
    std::vector<C2<R>> h(n), htw(4096);
    for (size_t i = 0; i < n; ++i) h[i] = {(R)((double)rand() / RAND_MAX - 0.5), (R)((double)rand() / RAND_MAX - 0.5)};
    for (int i = 0; i < 4096; ++i) htw[i] = {(R)cos(2 * M_PI * i / 4096), (R)-sin(2 * M_PI * i / 4096)};
    std::vector<double> hf(n, 1.0);
    HK(hipMemcpy(a, h.data(), n * sizeof(C2<R>), hipMemcpyHostToDevice));
    HK(hipMemcpy(tw, htw.data(), 4096 * sizeof(C2<R>), hipMemcpyHostToDevice));
    HK(hipMemcpy(filt, hf.data(), n * sizeof(double), hipMemcpyHostToDevice));
    const int tiles = (ncols + CT - 1) / CT;
    hipEvent_t e[4]; for (auto& x : e) HK(hipEventCreate(&x));
    auto report = [&](const char* name, float ms1, float ms2) {
        const double by = 2.0 * H * ncols * sizeof(C2<R>);
        printf("  CT=%d %-40s pass1 %.1f us  pass2 %.1f us  both %.1f us  = %.2f TB/s of one read + one write\n", CT, name, 1e3 * ms1 / reps,
               1e3 * ms2 / reps, 1e3 * (ms1 + ms2) / reps, by / 1e12 / ((ms1 + ms2) / reps / 1e3));
    };
    // variants: (A) pass 2 writes the natural row order, (B) pass 2 writes row 64 k1 + k2 (rows out of order, in place),
    // (C) pass 1 writes rows 64 r1 + k1, pass 2 reads them 64 apart and writes 64 k1 + k2; each with / without the filter product
    for (int variant = 0; variant < 5; ++variant) {
        float t1 = 0, t2 = 0;
        for (int i = -2; i < reps; ++i) {
            HK(hipEventRecord(e[0]));
            const dim3 g(tiles, 64), bl(CT * 8);
            switch (variant) {
            case 0: case 1: k_pass<R, CT, true, true, true, false><<<g, bl>>>(a, b, filt, tw, pitch, ncols, 0); break;
            case 2: k_pass<R, CT, true, false, true, false><<<g, bl>>>(a, b, filt, tw, pitch, ncols, 0); break;
            case 3: case 4: k_pass<R, CT, true, true, true, true><<<g, bl>>>(a, b, filt, tw, pitch, ncols, 0); break;
            }
            HK(hipEventRecord(e[1]));
            switch (variant) {
            case 0: case 3: k_pass<R, CT, false, true, false, false><<<g, bl>>>(b, c, filt, tw, pitch, ncols, 0); break;
            case 1: case 4: k_pass<R, CT, false, false, false, false><<<g, bl>>>(b, b, filt, tw, pitch, ncols, 0); break;
            case 2: k_pass<R, CT, true, false, false, false><<<g, bl>>>(b, c, filt, tw, pitch, ncols, 0); break;
            }
            HK(hipEventRecord(e[2]));
            HK(hipEventSynchronize(e[2]));
            float x, y; HK(hipEventElapsedTime(&x, e[0], e[1])); HK(hipEventElapsedTime(&y, e[1], e[2]));
            if (i >= 0) { t1 += x; t2 += y; }
        }
        const char* names[5] = {"A: 64-apart -> 64-apart, block -> natural", "B: 64-apart -> 64-apart, block -> block (in place)",
                                "C: 64-apart -> block, 64-apart -> block", "A + filter product at the load", "B + filter product at the load"};
        report(names[variant], t1, t2);
    }
    // halves of the columns one after the other (pass 1, pass 2 of the left half, then of the right): the intermediate of a half
    // (67 MB fp64) can stay in the Infinity Cache between its two passes
    {
        float t = 0;
        const int th = (tiles + 1) / 2;
        for (int i = -2; i < reps; ++i) {
            HK(hipEventRecord(e[0]));
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int t0 = hlf * th, nt = hlf ? tiles - th : th;
                k_pass<R, CT, true, true, true, true><<<dim3(nt, 64), CT * 8>>>(a, b, filt, tw, pitch, ncols, t0);
                k_pass<R, CT, false, true, false, false><<<dim3(nt, 64), CT * 8>>>(b, c, filt, tw, pitch, ncols, t0);
            }
            HK(hipEventRecord(e[1])); HK(hipEventSynchronize(e[1]));
            float x; HK(hipEventElapsedTime(&x, e[0], e[1]));
            if (i >= 0) t += x;
        }
        report("A + filter, column halves in turn", t, 0);
        t = 0;
        const int tq = (tiles + 3) / 4;
        for (int i = -2; i < reps; ++i) {
            HK(hipEventRecord(e[0]));
            for (int qt = 0; qt < 4; ++qt) {
                const int t0 = qt * tq, nt = (t0 + tq > tiles) ? tiles - t0 : tq;
                if (nt <= 0) break;
                k_pass<R, CT, true, true, true, true><<<dim3(nt, 64), CT * 8>>>(a, b, filt, tw, pitch, ncols, t0);
                k_pass<R, CT, false, true, false, false><<<dim3(nt, 64), CT * 8>>>(b, c, filt, tw, pitch, ncols, t0);
            }
            HK(hipEventRecord(e[1])); HK(hipEventSynchronize(e[1]));
            float x; HK(hipEventElapsedTime(&x, e[0], e[1]));
            if (i >= 0) t += x;
        }
        report("A + filter, column quarters in turn", t, 0);
    }
    // a plain copy of the same columns, for the box's own ceiling
    {
        float t = 0;
        for (int i = -2; i < reps; ++i) {
            HK(hipEventRecord(e[0]));
            k_copy<R><<<dim3((ncols + 255) / 256, H), 256>>>(a, c, pitch, ncols);
            HK(hipEventRecord(e[1])); HK(hipEventSynchronize(e[1]));
            float x; HK(hipEventElapsedTime(&x, e[0], e[1]));
            if (i >= 0) t += x;
        }
        report("copy of the same columns (one pass)", t, 0);
    }
    // check variant A against a direct sum on a few columns and rows
    k_pass<R, CT, true, true, true, false><<<dim3(tiles, 64), CT * 8>>>(a, b, filt, tw, pitch, ncols, 0);
    k_pass<R, CT, false, true, false, false><<<dim3(tiles, 64), CT * 8>>>(b, c, filt, tw, pitch, ncols, 0);
    HK(hipDeviceSynchronize());
    std::vector<C2<R>> r(n);
    HK(hipMemcpy(r.data(), c, n * sizeof(C2<R>), hipMemcpyDeviceToHost));
    double maxerr = 0;
    const int cols[4] = {0, 17, ncols / 2, ncols - 1}, ks[6] = {0, 1, 63, 64, 1234, 4095};
    for (int col : cols) for (int k : ks) {
        long double sr = 0, si = 0;
        for (int y = 0; y < H; ++y) {
            const long double ang = -2.0L * M_PIl * ((long long)y * k % 4096) / 4096.0L;
            const long double cr = cosl(ang), ci = sinl(ang);
            sr += h[(size_t)y * pitch + col].x * cr - h[(size_t)y * pitch + col].y * ci;
            si += h[(size_t)y * pitch + col].x * ci + h[(size_t)y * pitch + col].y * cr;
        }
        maxerr = fmax(maxerr, fmax(fabs((double)(sr - r[(size_t)k * pitch + col].x)), fabs((double)(si - r[(size_t)k * pitch + col].y))));
    }
    printf("  CT=%d variant A against the direct sum: max abs error %.3g\n", CT, maxerr);
    HK(hipFree(a)); HK(hipFree(b)); HK(hipFree(c)); HK(hipFree(tw)); HK(hipFree(filt));
    return 0;
}

int main(int argc, char** argv) {
    const int prec = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 20, ncols = argc > 3 ? atoi(argv[3]) : 2049;
    printf("two-pass column transform, H=4096, %d columns of a 4096-wide plane, %s\n", ncols, prec ? "f64" : "f32");
    if (prec) { run<double, 16>(reps, ncols); run<double, 32>(reps, ncols); run<double, 8>(reps, ncols); }
    else { run<float, 32>(reps, ncols); run<float, 64>(reps, ncols); run<float, 16>(reps, ncols); }
    return 0;
}

// Internal C++ declarations of the non-FFT kernels' host launchers.
#pragma once
#include "nf_common.h"

namespace nf {

struct WeightVec {
    double v[8];
};

// the scales of a tile as the batch dimension of one launch (nf_scale.hip): per-entry operands, by value
#define NF_SB_MAX 8
struct BatchPtrs {
    void* p[NF_SB_MAX];
};
struct FinalizeBatch {
    void* filtered[NF_SB_MAX];
    const float* rms[NF_SB_MAX];
    const unsigned char* edgeCheck[NF_SB_MAX];
    void* snOut[NF_SB_MAX];
    float* rmsOut[NF_SB_MAX];
    unsigned char* maskOut[NF_SB_MAX];
    unsigned long long* chunkRange[NF_SB_MAX];
    int* zeroRows[NF_SB_MAX];
};

// nf_pointwise.hip
template <typename T>
int cross_power(const void* F, void* NP, int N, long long npix, hipStream_t s, int halfH = 0, int halfW = 0);
size_t xmatch_workspace_bytes(int n);
int xmatch_self(const double* u, const double* snr, int n, double lo2, double hi2, int* count, int* best, unsigned char* drop,
                int* nearFlag, void* work, size_t workBytes, hipStream_t s);
#define NF_XM_STATIC_MAX_ROWS 16384
int xmatch_static_scratch(void** ptr);
int xmatch_self_static(const double* u, const double* snr, int n, double lo2, double hi2, int* count, int* best, unsigned char* drop,
                       int* nearFlag, hipStream_t s);
int symmetrise_filter(const float* filt, double* filtSym, int N, int H, int W, hipStream_t s);
int build_filter(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, float* filt, int N,
                 long long npix, hipStream_t s, int halfH = 0, int halfW = 0);
int build_filter_sym(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, float* filt, double* sym, int N,
                     int H, int W, hipStream_t s);
bool build_filter_shares_inverse(int nb, int N);     // the batch form inverts a mode's covariance once for all nb entries
int build_filter_sym_batch(int covType, int sigType, const void* cov, const void* S, const WeightVec& w, const BatchPtrs& filt, int nb,
                           double* sym, int N, int H, int W, hipStream_t s);
int finalize_batch(int vtype, const FinalizeBatch& fb, int nb, const unsigned char* surveyMask, const unsigned char* psMask, int H, int W,
                   int apodPix, int mulSurvey, hipStream_t s);
template <typename T>
int apply_filter(const void* F, const float* filt, void* G, int N, int sum, long long npix, hipStream_t s);
template <typename T>
int window_mul(void* F, const double* wy, const double* wx, int H, int W, int batch, hipStream_t s);
int paint_radial(float* out, int H, int W, const double* A, const double* B, const double* C, const double* r,
                 const double* prof, const double* slope, int ntab, double r0, double dlnr, double amp, double sign,
                 hipStream_t s);
struct PaintPlanes {       // per plane: the radial table as paint_radial takes it (log2r0 / invDlog2 as that function forms them)
    int n;
    const double* r[8];
    const double* prof[8];
    const double* slope[8];
    int ntab[8];
    float log2r0[8], invDlog2[8], sign[8];
    double amp[8];
};
int paint_radial_planes(float* out, int H, int W, const double* A, const double* B, const double* C, const PaintPlanes& P,
                        hipStream_t s);
int paint_radial2(float* out, float* out2, int H, int W, const double* A, const double* B, const double* C, const double* r,
                  const double* prof, const double* slope, const double* prof2, const double* slope2, int ntab, double r0,
                  double dlnr, double amp, double sign, double amp2, double sign2, hipStream_t s);
int scale_mask(int vtype, const float* in, void* out, double norm, const double* normDev, const unsigned char* psMask,
               long long n, hipStream_t s);
int finalize(int vtype, void* filtered, const float* rms, const unsigned char* surveyMask, const unsigned char* psMask,
             const unsigned char* edgeCheck, void* snOut, float* rmsOut, unsigned char* maskOut, int H, int W, int apodPix,
             int mulSurvey, unsigned long long* chunkRange, int* zeroRows, hipStream_t s);
int sum_planes(const float* in, float* out, int N, long long n, hipStream_t s);
int sn_cached(int vtype, const void* data, const float* rms, double* sn, long long n, hipStream_t s);
int restore_zeros(int vtype, void* data, const void* ref, long long n, hipStream_t s);

// nf_stencil.hip
int gaussian_filter(int rtype, const void* in, void* tmp, void* out, int H, int W, int batch, const double* wY, int radY,
                    const double* wX, int radX, hipStream_t s);
int subtract(int rtype, const void* a, const void* b, void* out, long long n, hipStream_t s);
int convolve2d(int rtype, const void* in, void* out, int H, int W, const double* kernFlipped, int kh, int kw, hipStream_t s);
int edge_check(int vtype, const void* filtered, const unsigned char* psMask, unsigned char* tmp, unsigned char* out, int H, int W,
               int size, hipStream_t s);

// nf_preproc.hip
int median_filter(int rtype, const void* in, void* out, int H, int W, int size, hipStream_t s);
int disc_fill(double* rmap, int W, const double* rec, const int* box, int boxPixels, hipStream_t s);
int disc_apply(int rtype, void* data, const void* bck, unsigned char* surveyMask, unsigned char* psMask, int W, const double* recs,
               const int* boxes, int nBoxes, double limit, int times60, const double* rmap, hipStream_t s);
int paint_catalog(float* out, int H, int W, int nObj, const double* obj, const int* win, const double* geom, const double* prof,
                  int ntab, int accumulate, hipStream_t s);
int add_model(int rtype, void* data, const float* model, unsigned char* flagMask, long long n, int subtract, hipStream_t s);
int load_tile(int rtype, const void* map, int wtype, const void* weights, const unsigned char* survey, const unsigned char* ps,
              const unsigned char* flag, long long pitch, int x0, int y0, int H, int W, int ax0, int ax1, int ay0, int ay1, void* dataOut,
              void* wOut, unsigned char* surveyOut, unsigned char* psOut, unsigned char* flagOut, hipStream_t s);
int inject_model(int baseType, const void* base, int modelType, const void* model, int wType, const void* weights, void* out, long long n,
                 hipStream_t s);
int finalize_cached(int vtype, void* filtered, const float* rms, const unsigned char* surveyMask, const unsigned char* psMask,
                    const unsigned char* edgeCheck, double* snOut, unsigned char* maskOut, int H, int W, int apodPix, int mulSurvey,
                    unsigned long long* chunkRange, int* zeroRows, hipStream_t s);

// nf_noise.hip
struct NoiseGrid {
    int ny, nx;          // number of cells per axis
    int overlap;         // pixels added on each side of a cell
    int numBins;
    int estimator;       // 0: 3-sigma clipped std, 1: 68.3 percentile of |v|
    double pctQ;         // 68.3 / 100 as numpy forms it (estimator 1)
    const int* yEdges;   // device, ny+1
    const int* xEdges;   // device, nx+1
    double quantiles[16];  // percentile/100 of the lower edge of each weight bin (np.arange(0,100,100/numBins)/100)
};
int noise_map(int vtype, const void* filtered, const void* weights, int wtype, float* rms, int H, int W, const NoiseGrid& g,
              void* cellStats, size_t cellStatsBytes, bool tablesValid, hipStream_t s);
bool noise_batchable(const NoiseGrid& g);
int noise_map_batch(int vtype, const void* filtered, long long vStrideBytes, const void* weights, int wtype, float* const* rmsList, int nb,
                    int nbTables, int H, int W, const NoiseGrid& g, void* cellStats, size_t cellStatsBytes, long long wsStrideBytes,
                    bool tablesValid, hipStream_t s);
size_t noise_workspace_bytes(const NoiseGrid& g, int H, int W);
size_t noise_smart_workspace_bytes();
int noise_map_smart(int vtype, const void* filtered, const void* weights, int wtype, float* rms, long long n, int numBins,
                    void* work, size_t workBytes, hipStream_t s);
int median_stack(int rtype, const void* in, void* out, int N, long long n, hipStream_t s);

// nf_detect.hip
long long label_chunk_flags_offset(int H, int W);
int label_components(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work,
                     int* numObjects, int* dirty, hipStream_t s);
int object_stats(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* stats,
                 const int* chunkFG, hipStream_t s);
int label_and_stats(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work, int* dirty,
                    int maxObjects, double* stats, const unsigned long long* snRange, hipStream_t s);
int label_and_stats_batch(int vtype, int nb, const void* const* sn, double threshold, int negate, int* const* labels, int H, int W,
                          int* const* work, int* const* dirty, int maxObjects, double* const* stats,
                          const unsigned long long* const* snRange, hipStream_t s);
int object_shapes(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* shapes, hipStream_t s);
int resample_bilinear(const float* in, float* out, int N, int H, int W, int H2, int W2, const double* yIdx, const double* xIdx,
                      hipStream_t s);
int rms_table(const float* rms, const unsigned char* areaMask, const int* flagMask, int maxFlags, const unsigned char* intersect,
              const unsigned long long* rowAreaFix, int H, int W, unsigned* keys, unsigned long long* cnt, unsigned long long* area,
              int tableSize, int* overflow, hipStream_t s);
int map_max(int vtype, const void* m, long long n, double* out, hipStream_t s, int nb = 1, long long mapStride = 0);
int sum_planes_max(const float* in, float* out, int N, long long n, double* mapMax, hipStream_t s, int nb = 1);
int map_max_finish_weights_batch(double* mapMax, long long n, const float* planes, int N, const BatchPtrs& relOut, const double* peak,
                                 double y0, const BatchPtrs& normOut, double* normWs, int nb, hipStream_t s);
int map_max_finish_weights(double* mapMax, long long n, const float* planes, int N, double* relOut, const double* peak, double y0,
                           double* normOut, hipStream_t s);
int ring_flags(const int* ringLabels, const double* ringStats, int numRing, int minRingPix, int H, int W, unsigned char* isRing,
               hipStream_t s);
int spline_eval(int vtype, const void* map, int negate, int H, int W, const double* yx, int nPts, int halfWin, int pointsClass, const int* edgeSel, int nEdge,
                double* out, hipStream_t s);

// jobs of one batched spline evaluation (kernel argument, by value)
#define NF_SPL_MAXJOBS 32
struct SplineBatch {
    const void* map[NF_SPL_MAXJOBS];      // the maps, all of one dtype
    const double* yx[NF_SPL_MAXJOBS];     // (y, x) pairs
    const int* edge[NF_SPL_MAXJOBS];      // indices of the points whose window meets the border (nullptr: try every point)
    double* out[NF_SPL_MAXJOBS];
    int n[NF_SPL_MAXJOBS], nEdge[NF_SPL_MAXJOBS], negate[NF_SPL_MAXJOBS];
};
int spline_eval_batch(int vtype, const SplineBatch& b, int nJobs, int H, int W, int halfWin, hipStream_t s);

}  // namespace nf

/*
 * nemofilt - C ABI of the MI355X (gfx950) map-filtering + detection kernels.
 *
 * The reference (simonsobs/nemo) is pure Python and has no FFI for this path: the seam is the Python class
 * protocol filters.MapFilter / photometry.findObjects.  This header is the boundary the
 * MI355X-native Python classes in nemo_amd/ bind with ctypes (INTEGRATION.md shows the stub); each entry
 * point cites the reference statement(s) it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - every pointer named *_d / documented "device" is a device (HBM) pointer owned by the caller; the library
 *    allocates only plan tables (nf_plan_create) and never frees caller memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it, and
 *    the call returns without synchronising unless stated;
 *  - maps are row-major [batch][H][W]; complex data is interleaved (re, im);
 *  - return value: 0 on success, negative NF_ERR_* otherwise, with a message available from nf_last_error();
 *    nothing is thrown across the boundary.
 *  - `prec`, `*type` arguments: NF_F32 (0) or NF_F64 (1).
 */
#ifndef NEMOFILT_H
#define NEMOFILT_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NF_F32 0
#define NF_F64 1

#define NF_IN_COMPLEX 0 /* complex of precision `prec`      */
#define NF_IN_REAL64 1  /* real float64 (promoted)          */
#define NF_IN_REAL32 2  /* real float32 (promoted)          */

typedef struct nf_plan nf_plan;

int nf_version(void);
const char* nf_last_error(void);
/* number of visible HIP devices, or negative error; selects nothing */
int nf_device_count(void);

/* FFT plan for one tile shape (twiddles, radix schedule, Bluestein chirps for lengths with a prime factor > 13,
 * e.g. the 1031 x 1032 test tile of tests/configs/smallTestSurveyMaskHeader.txt:4-5).  Tables for a precision are
 * built on first use.  The plan belongs to the device that is current when it is created. */
int nf_plan_create(int H, int W, nf_plan** plan);
int nf_plan_destroy(nf_plan* plan);

/* enmap.apply_window(m, pow) (pixell; called at nemo/filters.py:98-101 to undo the pixel window, nemo/maps.py:1849 to apply it)
 * multiplies the 2-D spectrum of a real map by wy[ky] * wx[kx]; the two 1-D operators commute, and nf_axis_filter is one of
 * them: y = IFFT(FFT(x) * w) along axis 0 (x, within rows) or 1 (y, within columns) of `batch` real [H, W] maps, in ONE pass
 * over the map (two real lines share one complex transform; the spectrum stays in LDS).  wperm_d: n doubles with
 * wperm[rev[k]] = w[k] / n, rev from nf_plan_digit_reversal (n = W for axis 0, H for axis 1); w must be real and even
 * (w[k] == w[n-k]).  zeroRef_d (optional, `batch` maps of the output type, laid out like the output): where map b of it is zero,
 * map b of the output is zero (nemo/filters.py:99-101); zeroRows_d (optional, with zeroRef_d, axis 0): [batch][H] ints, 0 = that
 * row of that map of zeroRef_d holds no zero - such rows are not looked up (nf_finalize makes the table).  in_d == out_d is allowed.  Lengths with a prime factor above 13 are refused (NF_ERR_SIZE,
 * *bluestein = 1 from nf_plan_digit_reversal): use the 2-D transforms there. */
int nf_plan_digit_reversal(nf_plan* plan, int prec, int axis, unsigned short* rev_host, int* bluestein);
int nf_axis_filter(nf_plan* plan, int prec, int axis, const void* in_d, int inType, void* out_d, int outType, int batch,
                   const double* wperm_d, const void* zeroRef_d, const int* zeroRows_d, void* stream);

/* enmap.fft (pixell), as used at nemo/filters.py:525,548,559,615,646,679,845: full complex 2-D DFT of
 * `batch` maps.  Input may be real (promoted) or complex.  If apodY_d/apodX_d are non-NULL the real input is
 * multiplied by the apodisation profiles first - enmap.apod, nemo/filters.py:525 - (axis 0 profile, then
 * axis 1, each product rounded to the input dtype).  unitary != 0 divides by sqrt(H*W) as enmap.fft does. */
int nf_fft2_forward(nf_plan* plan, int prec, const void* in_d, int inType, void* out_d, int batch,
                    const double* apodY_d, const double* apodX_d, int unitary, void* stream);

/* enmap.ifft (pixell), nemo/filters.py:662,847: backward DFT (plain sum; unitary != 0 divides by sqrt(H*W),
 * unitary == 0 is normalize=False).  `work_d` is a complex scratch of the same size as `in_d` (may alias in_d
 * if the input may be overwritten).  outType NF_IN_COMPLEX writes complex; NF_IN_REAL64 / NF_IN_REAL32 write
 * the real part only (np.real(...), and .astype(np.float32) for REAL32). */
int nf_fft2_inverse(nf_plan* plan, int prec, const void* in_d, void* work_d, void* out_d, int outType, int batch,
                    int unitary, void* stream);

/* nemo/filters.py:847 / :100,644 in one pass over the Fourier plane: the inverse transform of
 *     win * sum_i in_i * filt_i          (nSum > 1: batch must be 1, in_d / filt_d hold nSum planes)
 *     win * in_b * filt_b                (nSum == 1: per batch entry; filtPerBatch == 0 uses one filter plane for all)
 * where filt_d (float32, may be NULL = 1) is the matched filter and win = wy[y] * wx[x] (float64 tables of
 * enmap.apply_window already raised to the power wanted; both NULL = 1).  The products are formed while the column pass
 * loads its operands, so nf_apply_filter / nf_window_mul and their extra read + write of the plane are not needed.
 * Other arguments as nf_fft2_inverse. */
int nf_fft2_inverse_fused(nf_plan* plan, int prec, const void* in_d, const float* filt_d, int nSum, int filtPerBatch,
                          const double* wy_d, const double* wx_d, void* work_d, void* out_d, int outType, int batch,
                          int unitary, void* stream);

/* nemo/filters.py:573: NP[i][j] = Re(F_i conj F_j) for all i, j;  F [N][npix] complex, NP [N][N][npix] real. */
int nf_cross_power(int prec, const void* F_d, void* NP_d, int N, long long npix, void* stream);

/* ---- half-stored spectra -------------------------------------------------------------------------------------
 * Every transform on this path starts from or ends in a REAL map (nemo/filters.py:525,615,646,847), whose spectrum obeys
 * F[-y][-x] = conj F[y][x].  The *_half entry points keep the [H][W] plane layout of the full-spectrum calls but store
 * and read only the columns x <= W/2; the rest of each plane is never written nor read (its content is undefined).  The
 * column pass of each transform then moves half of the bytes.  Values are those of the full-spectrum calls: the missing
 * modes are used through the symmetry, in the same arithmetic.
 *   nf_fft2_forward_half  = nf_fft2_forward of real input (NF_IN_REAL64 / NF_IN_REAL32), columns x <= W/2 only
 *   nf_symmetrise_filter  filtSym[k] = 0.5 (filt[k] + filt[-k]) in float64, half-stored: the smoothed noise power of
 *                           nemo/filters.py:580 is not exactly symmetric under k -> -k, so neither is the filter, and
 *                           real(ifft2(F * filt)) = ifft2(F * filtSym) for the spectrum F of a real map
 *   nf_fft2_inverse_half  = nf_fft2_inverse_fused with real output of half-stored input planes and the symmetrised filter
 *                           (NULL = none); the window tables must be even, wy[H-y] == wy[y], as enmap.apply_window's
 *                           sinc(fftfreq) tables are.  planeScale (HOST doubles, or NULL): one more real factor per plane of
 *                           the sum (nSum > 1: nSum entries) or per batch entry (nSum == 1: batch entries), at most 16 -
 *                           the amplitude of a template whose spectrum is given for unit amplitude (nemo/filters.py:639-646).
 *                           rowLo < rowHi: only the output rows rowLo .. rowHi - 1 (widened to whole row pairs) are
 *                           computed, the rest of out_d is left as it was - nemo/filters.py:654-657 reads the filtered
 *                           template only around the tile centre; both 0: the whole map
 *   nf_cross_power_half, nf_build_filter_half = nf_cross_power / nf_build_filter reading half-stored F_d / S_d and writing
 *                           full planes */
int nf_fft2_forward_half(nf_plan* plan, int prec, const void* in_d, int inType, void* out_d, int batch,
                         const double* apodY_d, const double* apodX_d, int unitary, void* stream);
int nf_symmetrise_filter(const float* filt_d, double* filtSym_d, int N, int H, int W, void* stream);
int nf_fft2_inverse_half(nf_plan* plan, int prec, const void* in_d, const double* filtSym_d, int nSum, int filtPerBatch,
                         const double* wy_d, const double* wx_d, const double* planeScale, void* work_d, void* out_d,
                         int outType, int batch, int rowLo, int rowHi, int unitary, void* stream);
/* nf_fft2_inverse_half of ONE map (nSum planes summed, or one plane) with the pointwise steps that follow it on the path
 * applied while the real rows are stored (they would otherwise re-read and re-write the map):
 *   scale_d (device scalar) [+ mask_d]: out = float32(value) * scale [* mask], in the output type - the cast of
 *       nemo/filters.py:847, the signalNorm product of :853 and the point-source mask of :714 (= nf_scale_mask_dev);
 *   zeroRef_d (map of the output type): out = 0 where zeroRef is 0 - nemo/filters.py:99-101 (= nf_restore_zeros).
 * Either may be NULL.  Other arguments as nf_fft2_inverse_half. */
int nf_fft2_inverse_half_epi(nf_plan* plan, int prec, const void* in_d, const double* filtSym_d, int nSum, const double* wy_d,
                             const double* wx_d, void* work_d, void* out_d, int outType, int unitary, const double* scale_d,
                             const unsigned char* mask_d, const void* zeroRef_d, void* stream);
int nf_cross_power_half(int prec, const void* F_d, void* NP_d, int N, int H, int W, void* stream);
int nf_build_filter_half(int covType, int sigType, const void* cov_d, const void* S_d, const double* w, float* filt_d,
                         int N, int H, int W, void* stream);
/* nf_build_filter_half and nf_symmetrise_filter in one pass: the same filter cube filt_d and the same symmetrised form
 * filtSym_d, bit for bit, without reading the cube back (a thread computes a mode and its mirror mode together).  filt_d may
 * be NULL: a caller that only applies the filter (and neither saves nor returns the cube) gets the symmetrised form alone. */
int nf_build_filter_sym_half(int covType, int sigType, const void* cov_d, const void* S_d, const double* w, float* filt_d,
                             double* filtSym_d, int N, int H, int W, void* stream);

/* scipy.ndimage.gaussian_filter with mode='reflect' (nemo/filters.py:580 with sigma (3,3); nemo/maps.py:1380).
 * wY_d / wX_d: normalised 1-D kernels of length 2*rad+1 (device, float64) for axis 0 / axis 1, as
 * scipy builds them; tmp_d holds the axis-0 result in the map dtype. */
int nf_gaussian_filter(int rtype, const void* in_d, void* tmp_d, void* out_d, int H, int W, int batch,
                       const double* wY_d, int radY, const double* wX_d, int radX, void* stream);

/* nemo/filters.py:621-627: filt[:, k] = inv(noiseCov[:, :, k]) . (w * |fSignal[:, k]|), float32; singular modes
 * stay 0.  cov_d [N][N][npix] real (covType), S_d [N][npix] complex (sigType), w: N host doubles. */
int nf_build_filter(int covType, int sigType, const void* cov_d, const void* S_d, const double* w, float* filt_d,
                    int N, long long npix, void* stream);

/* nemo/filters.py:847 (product with the filter).  sum != 0: G[npix] = sum_i F_i * filt_i (the frequency sum is
 * moved in front of the inverse FFT, which is linear); sum == 0: G[N][npix] = F_i * filt_i per frequency
 * (nemo/filters.py:662, for fRelWeights). */
int nf_apply_filter(int prec, const void* F_d, const float* filt_d, void* G_d, int N, int sum, long long npix,
                    void* stream);

/* nemo/filters.py:847, the frequency sum AFTER the inverse transforms, as the reference takes it: out_d[n] = in_d[0][n] +
 * in_d[1][n] + ... in float32, planes added in order (ndarray.sum(axis=0)). */
int nf_sum_planes(const float* in_d, float* out_d, int N, long long n, void* stream);

/* pixel-window multiplication inside enmap.apply_window (nemo/filters.py:100,644): F[b][y][x] *= wy[y]*wx[x]
 * with wy = sinc(fftfreq(H))**pow, wx likewise (float64 device tables). */
int nf_window_mul(int prec, void* F_d, const double* wy_d, const double* wx_d, int H, int W, int batch, void* stream);

/* Template painter standing where pixell.pointsrcs.sim_objects stands (nemo/signals.py:681,691): float32 map of
 * sign * float32(amp * profile(theta)); hav(theta) = A[y] + B[y]*C[x] (device tables), profile by linear
 * interpolation on the log-uniform table r_d/prof_d (ntab entries, r[k] = r0*exp(k*dlnr)).  slope_d (optional, may be
 * NULL): ntab-1 doubles (prof[k+1]-prof[k])/(r[k+1]-r[k]), which saves the kernel one float64 division per pixel. */
int nf_paint_radial(float* out_d, int H, int W, const double* A_d, const double* B_d, const double* C_d,
                    const double* r_d, const double* prof_d, const double* slope_d, int ntab, double r0, double dlnr,
                    double amp, double sign, void* stream);

/* Two templates on the same radii in one pass (the unit template of nemo/filters.py:611-618 and the amplitude-scaled one
 * of :639-646 are painted at the same place: angle and table search are shared); each output as nf_paint_radial gives it. */
int nf_paint_radial2(float* out_d, float* out2_d, int H, int W, const double* A_d, const double* B_d, const double* C_d,
                     const double* r_d, const double* prof_d, const double* slope_d, const double* prof2_d,
                     const double* slope2_d, int ntab, double r0, double dlnr, double amp, double sign, double amp2,
                     double sign2, void* stream);

/* nemo/filters.py:853 + :714: out = float32 map * signalNorm * psMask, in float64 (vtype NF_F64: signalNorm is a
 * float64 scalar, 'yc') or float32 (vtype NF_F32: 'uK').  psMask_d may be NULL. */
int nf_scale_mask(int vtype, const float* in_d, void* out_d, double signalNorm, const unsigned char* psMask_d,
                  long long n, void* stream);
/* The same with the normalisation read from device memory (one float64; for NF_F32 maps it holds the float32-rounded
 * value): lets the caller keep signalNorm = y0 / peak (nemo/filters.py:659) or 1 / max (:684) on the device and go on
 * enqueueing work without waiting for it. */
int nf_scale_mask_dev(int vtype, const float* in_d, void* out_d, const double* signalNorm_d, const unsigned char* psMask_d,
                      long long n, void* stream);

/* MapFilter.makeNoiseMap, grid mode (nemo/filters.py:407-478): local clipped RMS in ny x nx cells (edges as
 * np.linspace(..., dtype=int), device int arrays of ny+1 / nx+1 entries), each widened by `overlap` pixels,
 * with `numBins` weight-percentile bins per cell (quantiles: numBins host doubles = percentiles/100).
 * weights_d: the median weight map (float32/float64).  work_d: nf_noise_workspace_bytes() bytes of device scratch (for the
 * default estimator with one bin this includes the per-block sums, the tail lists in list and in magnitude-class
 * order and the class tables of the single-read evaluation, about 16 bytes per pixel).
 * estimator: 0 = 3-sigma clipped standard deviation (10 iterations, the default), 1 = RMSEstimator 'percentile'
 * (np.percentile(abs(values), 68.3), nemo/filters.py:459-460).  'biweight' needs astropy.stats and is not provided.
 * tablesValid (a hint that changes no result; 0 = never wrong): non-zero says that work_d was last used by a call of this entry
 * point with the same H, W, grid (ny, nx, edges, overlap), estimator, numBins = 1 and the same values in weights_d, and has not
 * been written since - the block tables it still holds (the cuts of the grid into blocks, every block's weight range) are
 * then not rebuilt and the weights are read only for blocks whose pixels are valid in part: the scales of one tile. */
size_t nf_noise_workspace_bytes(int H, int W, int ny, int nx, int numBins);
int nf_noise_map(int vtype, const void* filtered_d, const void* weights_d, int wtype, float* rms_d, int H, int W,
                 int ny, int nx, const int* yEdges_d, const int* xEdges_d, int overlap, int numBins,
                 const double* quantiles, int estimator, void* work_d, size_t workBytes, int tablesValid, void* stream);
/* MapFilter.makeNoiseMap with noiseGridArcmin: "smart" (nemo/filters.py:363-404): numBins edges
 * np.linspace(w.min(), w.max(), numBins) over the median weight map, per bin (strictly between two edges) the 3-sigma
 * clipped standard deviation of the non-zero map values, written to every pixel of the bin when positive.  n = H*W;
 * work_d: nf_noise_smart_workspace_bytes() bytes of device scratch.  Only the default estimator. */
size_t nf_noise_smart_workspace_bytes(void);
int nf_noise_map_smart(int vtype, const void* filtered_d, const void* weights_d, int wtype, float* rms_d, long long n,
                       int numBins, void* work_d, size_t workBytes, void* stream);
/* np.median over the frequency axis of N weight maps (nemo/filters.py:357-360); in_d [N][n] -> out_d [n]. */
int nf_median_stack(int rtype, const void* in_d, void* out_d, int N, long long n, void* stream);

/* ndimage.rank_filter(abs(m + (1 - psMask)), 0, size=(t, t)) > 0  (nemo/filters.py:735-736, 1139-1142);
 * tmp_d, out_d: H*W bytes.  psMask_d may be NULL. */
int nf_edge_check(int vtype, const void* filtered_d, const unsigned char* psMask_d, unsigned char* tmp_d,
                  unsigned char* out_d, int H, int W, int size, void* stream);

/* Tail of buildAndApply (nemo/filters.py:717-756 and :1112-1157): S/N map, survey/point-source/edge/apodisation
 * masks.  filtered_d is updated in place; mulSurvey != 0 for the Fourier-space variant (nemo/filters.py:741).
 * Mask inputs may be NULL (= all ones).  rmsOut_d may be rms_d (in place).  Values that do not change are not stored again. */
int nf_finalize(int vtype, void* filtered_d, const float* rms_d, const unsigned char* surveyMask_d,
                const unsigned char* psMask_d, const unsigned char* edgeCheck_d, void* sn_d, float* rmsOut_d,
                unsigned char* maskOut_d, int H, int W, int apodPix, int mulSurvey, unsigned long long* snChunkRange_d,
                int* zeroRows_d, void* stream);
/* snChunkRange_d (optional, NULL = not wanted): 2 * nf_sn_chunks(H, W) 64-bit words - for every chunk of 1024 consecutive
 * pixels (raster order) the largest S/N value written to sn_d, then, in the second half, the smallest (as order-preserving
 * integer keys).  Handed to nf_label_and_stats with the same sn_d, it lets the labelling pass skip, unread, every chunk whose
 * values cannot exceed the detection threshold - at 4 sigma nearly all of them (no counterpart in the reference: ndimage.label
 * reads the whole map).
 * zeroRows_d (optional): H ints, set to 1 for the rows of the updated filtered_d that hold a zero, else 0 - the table
 * nf_axis_filter takes beside its zero reference. */
int nf_sn_chunks(int H, int W);

/* nemo/pipelines.py:244-247 (useCachedRMSMap branch of the driver): sn_d (float64) = data, divided by the cached RMS map
 * (float32) where that is positive. */
int nf_sn_cached(int vtype, const void* data_d, const float* rms_d, double* sn_d, long long n, void* stream);

/* The tail of buildAndApply followed by the driver's cached-noise branch, in one pass (nemo/filters.py:725-748 then
 * nemo/pipelines.py:243-247; the re-filters of maps.sourceInjectionTest, nemo/maps.py:2047-2049): filtered_d is masked in place
 * as nf_finalize masks it, maskOut_d is nf_finalize's, and sn_d (float64) = the masked filtered map divided by the cached
 * noise map cachedRms_d (float32, not modified) where that is positive.  This pass's own noise map - which the reference
 * computes (nemo/filters.py:717) and the driver then drops - is not made.  snChunkRange_d / zeroRows_d as for nf_finalize. */
int nf_finalize_cached(int vtype, void* filtered_d, const float* cachedRms_d, const unsigned char* surveyMask_d,
                       const unsigned char* psMask_d, const unsigned char* edgeCheck_d, double* sn_d, unsigned char* maskOut_d,
                       int H, int W, int apodPix, int mulSurvey, unsigned long long* snChunkRange_d, int* zeroRows_d, void* stream);

/* nemo/filters.py:99-101: data[ref == 0] = 0 after the pixel-window deconvolution. */
int nf_restore_zeros(int vtype, void* data_d, const void* ref_d, long long n, void* stream);

/* nemo/maps.py:1290: out = a - b. */
int nf_subtract(int rtype, const void* a_d, const void* b_d, void* out_d, long long n, void* stream);

/* scipy.ndimage.convolve(map_float32, kern2d) with mode='reflect' (nemo/filters.py:1193).  kernFlipped_d is the
 * kernel already flipped on both axes (float64, kh x kw, odd sizes). */
int nf_convolve2d(const float* in_d, float* out_d, int H, int W, const double* kernFlipped_d, int kh, int kw,
                  void* stream);
/* The same for a float64 map (nemo/filters.py:1189-1193 with bckSub False: the float32 zeros + a float64 cube promote to
 * float64, so ndimage.convolve runs and returns in float64). */
int nf_convolve2d_f64(const double* in_d, double* out_d, int H, int W, const double* kernFlipped_d, int kh, int kw,
                      void* stream);

/* ---- maps.MapDict.preprocess on the device (nemo/maps.py:178-480) ------------------------------------------------------
 * scipy.ndimage.median_filter(data, size) with mode='reflect' (nemo/maps.py:366-368: the 20 x 20 moving median used as the
 * hole-filling background): window [i - size//2, i + size - size//2 - 1] on both axes, element of rank size*size//2.
 * size <= 48; in_d != out_d; rtype = map dtype. */
int nf_median_filter(int rtype, const void* in_d, void* out_d, int H, int W, int size, void* stream);

/* maps.makeDegreesDistanceMap (nemo/maps.py:2479-2536) and the masking / filling statements around its call sites
 * (:419-424 maskPointSourcesFromCatalog, :459-465 maskAndFillFromCatalog).  A hole is rec = {x0, y0, xPixScale, yPixScale,
 * limit} (device, 5 doubles) with box = {minX, maxX, minY, maxY} (device ints; boxHost / boxesHost: the same numbers on the host,
 * checked against the map before anything is launched).
 *   nf_disc_fill   rmap[y][x] = sqrt(((y - y0) yPixScale)^2 + ((x - x0) xPixScale)^2) inside the box - the persistent
 *                  distance map of :391, which the reference never resets between catalogue rows;
 *   nf_disc_apply  inside each of nBoxes boxes: where the distance (times 60 if times60) is below the limit, surveyMask = 0,
 *                  psMask = 0, data = bck.  rmap_d == NULL: distances are formed in place and compared with each hole's own
 *                  rec[4] (:459-465); else they are read from rmap_d and compared with `limit` (:422-424). */
int nf_disc_fill(double* rmap_d, int H, int W, const double* rec_d, const int* box_d, const int* boxHost, void* stream);
int nf_disc_apply(int rtype, void* data_d, const void* bck_d, unsigned char* surveyMask_d, unsigned char* psMask_d, int H, int W,
                  const double* recs_d, const int* boxes_d, const int* boxesHost, int nBoxes, double limit, int times60,
                  const double* rmap_d, void* stream);

/* The painting loop of maps.makeModelImage (nemo/maps.py:1790-1840 -> signals._paintSignalMap, nemo/signals.py:675-688):
 * out[y][x] (+)= sum over the nObj catalogue objects, IN ORDER, of the float32 value of object k's radial template at that
 * pixel, inside object k's window only (painter semantics of nf_paint_radial).  Per object: obj_d[k] = {amp, sign, f1, f2,
 * profile id, nf, -, -} (8 doubles; when nf > 0 the float32 value is multiplied by float32(f1), then float32(f2):
 * maps.convertToDeltaT on the painted clip, :684-686), win_d[k] = {x0, x1, y0, y1, offAB, offC, profile id, -} (8 ints; the
 * haversine tables of the window sit in geom_d: A then B per window row at offAB, C per window column at offC).  prof_d:
 * nProf tables of 3 * ntab doubles (radii, profile, slopes).  winHost: win_d on the host (bounds are checked first).
 * accumulate != 0 adds to what out_d holds. */
int nf_paint_catalog(float* out_d, int H, int W, int nObj, const double* obj_d, const int* win_d, const int* winHost,
                     const double* geom_d, long long geomLen, const double* prof_d, int nProf, int ntab, int accumulate,
                     void* stream);

/* MatchedFilter.reshapeFilter, nemo/filters.py:806-815: N float32 planes [H, W] sampled by linear interpolation at the
 * fractional source positions yIdx_d[H2] x xIdx_d[W2] (outside the grid: the edge value) -> out_d [N, H2, W2]. */
int nf_resample_bilinear(const float* in_d, float* out_d, int N, int H, int W, int H2, int W2, const double* yIdx_d,
                         const double* xIdx_d, void* stream);

/* MapDict.loadTile of a frequency map, its weights and its three masks plus the statements of MapDict.preprocess that follow
 * them (nemo/maps.py:83-175, 232-262) for maps resident in HBM, in one pass: the H x W clip at (y0, x0) of planes of srcH x srcW
 * pixels; data = map with 0 where weights == 0 (:243); weights (NULL: 1 everywhere, written in wtype); surveyMask zeroed
 * outside rows ay0..ay1-1 / columns ax0..ax1-1 of the clip (the tile's areaMaskInClipSection, :159-164; ax0 < 0: no such
 * section), or (weights != 0) when there is no survey mask; pointSourceMask (NULL: 1); flagMask * surveyMask (NULL: 0,
 * :259-262).  Masks are uint8. */
int nf_load_tile(int rtype, const void* map_d, int wtype, const void* weights_d, const unsigned char* surveyMask_d,
                 const unsigned char* psMask_d, const unsigned char* flagMask_d, int srcH, int srcW, int x0, int y0, int H, int W,
                 int ax0, int ax1, int ay0, int ay1, void* dataOut_d, void* weightsOut_d, unsigned char* surveyOut_d,
                 unsigned char* psOut_d, unsigned char* flagOut_d, void* stream);

/* Source injection, nemo/maps.py:341-344: out = base + (weights == 0 ? 0 : model), n pixels; base / model float32 or float64
 * (NF_F32 / NF_F64), out_d float32 when both are float32, else float64 (numpy's promotion); weights_d (wType) may be NULL;
 * base_d is not modified (out_d may be base_d when the types agree). */
int nf_inject_model(int baseType, const void* base_d, int modelType, const void* model_d, int wType, const void* weights_d,
                    void* out_d, long long n, void* stream);

/* nemo/maps.py:435-437 (subtract != 0: data -= model, flagMask += (model > 1); flagMask_d may be NULL) and :344 (subtract == 0:
 * data += model). */
int nf_add_model(int rtype, void* data_d, const float* model_d, unsigned char* flagMask_d, long long n, int subtract,
                 void* stream);

/* photometry.getObjectPositions, nemo/photometry.py:235-241: labels_d[H*W] int32 = scipy.ndimage.label numbering
 * of (map > threshold) (negate != 0: of (-map > threshold), nemo/photometry.py:54-55); work_d: int32 scratch of
 * H*W + 2*(H*W/1024) + 4 entries.  Synchronises the stream and returns the number of objects in *numObjects. */
int nf_label_components(int vtype, const void* sn_d, double threshold, int negate, int* labels_d, int H, int W,
                        int* work_d, int* numObjects, void* stream);

/* nemo/photometry.py:243-246: per label id 0..numObjects a record of 6 doubles {count, sum v, sum v*y, sum v*x,
 * max v, raster index of first max}; stats_d must hold (numObjects+1)*NF_OBJECT_STATS_DOUBLES doubles (the tail is scratch).
 * The three sums are accumulated exactly, as integers of v * 2^40 (v clamped to [0, 8e6]): their values do not depend on the
 * order of the additions, so centre-of-mass positions are bit-reproducible from run to run (within 1e-12 of scipy's
 * sequential float64 sums). */
#define NF_OBJECT_STATS_DOUBLES 17
int nf_object_stats(int vtype, const void* sn_d, int negate, const int* labels_d, int H, int W, int numObjects,
                    double* stats_d, void* stream);

/* The two calls above for a caller that labels map after map of one shape (nemo/pipelines.py:267-289: every filter scale of
 * every tile) and at a threshold where almost no pixel is foreground:
 *   nf_label_components_kept   labels_d is KEPT by the caller between calls, together with dirty_d (one int per 1024-pixel
 *       chunk, nf_div_up(H*W, 1024) of them; both zero before the first call): the call clears only the chunks the previous
 *       call wrote and writes only the chunks that hold foreground now - same labels as nf_label_components;
 *   nf_object_stats_chunks     chunkFlags_d = work_d + nf_label_chunk_flags_offset(H, W) of the labelling call that made
 *       labels_d: chunks without foreground are skipped after one flag - same records as nf_object_stats. */
long long nf_label_chunk_flags_offset(int H, int W);
int nf_label_components_kept(int vtype, const void* sn_d, double threshold, int negate, int* labels_d, int H, int W,
                             int* work_d, int* dirty_d, int* numObjects, void* stream);
int nf_object_stats_chunks(int vtype, const void* sn_d, int negate, const int* labels_d, int H, int W, int numObjects,
                           double* stats_d, const int* chunkFlags_d, void* stream);

/* nf_label_components(_kept) and nf_object_stats_chunks as ONE enqueue without the host round trip for the object count: the
 * record table stats_d has room for maxObjects labels ((maxObjects + 1) * NF_OBJECT_STATS_DOUBLES doubles, records as
 * nf_object_stats writes them); the number of labels found stays on the device and comes back in stats_d[0] (the count field
 * of the background record, which is otherwise zero).  If it exceeds maxObjects the records are incomplete (labels above
 * maxObjects were skipped): repeat with room for that many.  dirty_d as nf_label_components_kept (NULL: labels_d is written
 * everywhere).  snChunkRange_d: the table nf_finalize made for this very sn_d, or NULL.  The call does not wait for the GPU: several maps' detections can be enqueued and their records fetched in
 * one transfer (photometry.findObjectsGroup). */
int nf_label_and_stats(int vtype, const void* sn_d, double threshold, int negate, int* labels_d, int H, int W, int* work_d,
                       int* dirty_d, int maxObjects, double* stats_d, const unsigned long long* snChunkRange_d, void* stream);

/* nf_label_and_stats for numMaps maps of one shape (the scales of a tile, nemo/pipelines.py:267-289) with every pass ONE launch
 * for all of them (groups of 8 maps): sn_d, labels_d, work_d, stats_d - and dirty_d, snChunkRange_d, each NULL or an array whose
 * entries may be NULL - are HOST arrays of numMaps device pointers, each as in nf_label_and_stats; no two maps may share a
 * labels, work, dirty or record buffer.  Same labels and records per map as numMaps single calls. */
int nf_label_and_stats_batch(int vtype, int numMaps, const void* const* sn_d, double threshold, int negate, int* const* labels_d,
                             int H, int W, int* const* work_d, int* const* dirty_d, int maxObjects, double* const* stats_d,
                             const unsigned long long* const* snChunkRange_d, void* stream);

/* The segment moments behind measureShapes (photometry.findObjects, nemo/photometry.py:136-150): per label id 0..numObjects
 * a record of 10 doubles {xMin, yMin, xMax, yMax, S, Sx, Sy, Sxx, Syy, Sxy} with S = sum v, Sx = sum v*(x - xMin),
 * Sxx = sum v*(x - xMin)^2 ... over the segment's pixels (v as in nf_object_stats: exact integer sums of v * 2^40).
 * shapes_d must hold (numObjects+1)*NF_OBJECT_SHAPES_DOUBLES doubles (the tail is scratch).  The ellipse itself
 * (nemo/photometry.py:151-184) is a few scalar operations per object and stays with the caller. */
#define NF_OBJECT_SHAPES_DOUBLES 30
int nf_object_shapes(int vtype, const void* sn_d, int negate, const int* labels_d, int H, int W, int numObjects,
                     double* shapes_d, void* stream);

/* completeness.getRMSTab, nemo/completeness.py:1319-1333: the distinct non-zero values of a float32 noise map and, per value,
 * the number of pixels and the survey area.  areaMask_d (uint8, optional: weight of a pixel's area, normally 0 / 1),
 * flagMask_d (int32, optional: pixels with flag > maxFlags carry no area, :1319-1321), intersect_d (uint8, optional: where 0
 * the noise map counts as 0 and the pixel carries no area, :1324-1327).  rowAreaFix_d: H uint64, the area of a pixel of row y
 * in units of 2^-50 deg^2 (maps.getPixelAreaArcmin2Map's per-row float32 values / 3600).  Result: an open-addressing hash
 * table of tableSize (power of two) slots - keys_d the float32 bits (0 = empty slot), counts_d, areas_d (2^-50 deg^2);
 * *overflow_d != 0: the table was too small, ask again with a larger one.  Sums are integer: independent of the order. */
int nf_rms_table(const float* rms_d, const unsigned char* areaMask_d, const int* flagMask_d, int maxFlags,
                 const unsigned char* intersect_d, const unsigned long long* rowAreaFix_d, int H, int W, unsigned* keys_d,
                 unsigned long long* counts_d, unsigned long long* areas_d, int tableSize, int* overflow_d, void* stream);

/* ndarray.max() / np.argmax of a whole map (nemo/filters.py:661,684; nemo/filters.py:1017-1019,1198): out_d holds
 * NF_MAP_MAX_DOUBLES doubles - [0] maximum, [1] flat index of its first occurrence, the rest is scratch (per-workgroup
 * partial results). */
#define NF_MAP_MAX_DOUBLES (4 + 2 * 2048)
int nf_map_max(int vtype, const void* map_d, long long n, double* out_d, void* stream);

/* catalogs.makeOptimalCatalog, method 'ref-forced' (nemo/catalogs.py:319-362): cross match of the stacked per-scale
 * catalogues with themselves.  unit_d [n][3] unit vectors of (RA, dec), snr_d [n].  Partners = rows closer than radiusDeg
 * (chord test; a row is its own partner).  count_d[i] = partners of row i; best_d[i] = the first partner with the highest
 * SNR (np.argmax over the partner list, first partner if any SNR is NaN); drop_d[j] = 1 if a partner with several partners
 * prefers another row to j (the rows the reference removes).  *nearFlag_d = 1 if any chord lies within 1e-9 of the limit:
 * the caller must then decide with the exact great-circle formula (the host implementation does).  work_d:
 * nf_xmatch_workspace_bytes(n) bytes of device scratch (per-slice partial results). */
size_t nf_xmatch_workspace_bytes(int n);
int nf_xmatch_self(const double* unit_d, const double* snr_d, int n, double radiusDeg, int* count_d, int* best_d,
                   unsigned char* drop_d, int* nearFlag_d, void* work_d, size_t workBytes, void* stream);

/* nf_xmatch_self for callers that cannot allocate.  It has NO counterpart in the reference: it is for callers that cannot
 * call hipMalloc - code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same kernels and
 * results as nf_xmatch_self, but the per-slice partial results live in a file-scope device array of the library (6 MiB on
 * each GPU the library runs on, held until the process ends) instead of a caller workspace, so n is limited to
 * nf_xmatch_static_max_rows() = 16384 rows; more is refused with NF_ERR_SIZE.  The call allocates nothing.  Every call on
 * one GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race - serialise them.
 * The array's address on a GPU is looked up once per device, by nf_plan_create (any plan) made with that device current - make
 * one before capturing; a call on a device that never saw nf_plan_create does the look-up itself (not capture-safe). */
int nf_xmatch_static_max_rows(void);
int nf_xmatch_self_static(const double* unit_d, const double* snr_d, int n, double radiusDeg, int* count_d, int* best_d,
                          unsigned char* drop_d, int* nearFlag_d, void* stream);

/* nemo/photometry.py:69-73: isRing_d[id] = 1 if component id has > minRingPix pixels and its centre of mass is
 * not one of its own pixels. */
int nf_ring_flags(const int* ringLabels_d, const double* ringStats_d, int numRing, int minRingPix, int H, int W,
                  unsigned char* isRing_d, void* stream);

/* RectBivariateSpline(kx=3, ky=3)(y, x) (nemo/photometry.py:82-84,128; nemo/filters.py:654-657) at nPts points;
 * yx_d: nPts pairs (y, x), out_d: nPts doubles; halfWin <= 34 (32 reproduces the whole-map fit to 1e-16).
 * Points further than halfWin + 2 pixels from every edge are evaluated as one weighted sum over the window (Green's
 * function of the spline's tridiagonal system), the others by a windowed not-a-knot solve that honours the map border.
 * pointsClass: 0 = unknown (both kernels are launched, each skips the other's points), 1 = the caller guarantees that
 * all points are interior, 2 = that none is - saves the launch that would do nothing.  With pointsClass 0 the caller may
 * name the non-interior points (edgeSel_d: nEdge indices into yx_d, or NULL): the windowed solve, whose workgroups hold
 * 85 KB of LDS each, is then launched for those only. */
int nf_spline_eval(int vtype, const void* map_d, int negate, int H, int W, const double* yx_d, int nPts, int halfWin,
                   int pointsClass, const int* edgeSel_d, int nEdge, double* out_d, void* stream);

/* nf_spline_eval for several (map, point list) jobs at once - what one detection pass reads off its maps at the object
 * positions (nemo/photometry.py:128 on the S/N map; :276-281 and :354-361 on the filtered map and on the reference scale's two
 * maps, called from nemo/pipelines.py:299-311): one launch per kernel instead of one per map.  All maps have dtype `vtype`
 * and shape [H, W].  Host arrays of nJobs (<= 32) entries: maps_d, negate, yx_d (nPts[j] pairs each), out_d (nPts[j] doubles
 * each), edgeSel_d / nEdge (indices of job j's points whose window meets the map border; edgeSel_d[j] NULL: unknown, every
 * point is offered to both kernels). */
int nf_spline_eval_batch(int vtype, int nJobs, const void* const* maps_d, const int* negate, const double* const* yx_d,
                         const int* nPts, const int* const* edgeSel_d, const int* nEdge, double* const* out_d, int H, int W,
                         int halfWin, void* stream);

/* One matched-filter scale of a tile from template to filtered map, enqueued by ONE call: the body of
 * MatchedFilter.buildAndApply (nemo/filters.py:521-777, noise method 'dataMap', no background subtraction) and the pixel-window
 * undo of filterMaps (nemo/filters.py:98-101), expressed as the entry points above in the order the reference runs its
 * statements.  The reference (and this library's Python host layer) walks that sequence statement by statement from the
 * interpreter, once per scale; on 1-2 Mpix tiles the interpreter then costs more than the kernels do.  Same kernels, same
 * results bit for bit (tests/test_gpu_scale_entry.py).
 *
 * nf_mf_tile: what the scales of one tile share.  nf_mf_scale: the radial tables of one scale's templates and its outputs.
 *   outputUnits 0 ('yc', nemo/filters.py:632-668): per map the unit template is painted, all N planes transformed at once,
 *       filter built (:621-627) and symmetrised.  The y0-scaled templates of the normalisation (:639-646) are the unit
 *       templates times templateScale[i] (same profile, another amplitude: equal up to the float32 rounding of the painted
 *       pixels, 6e-8), so their spectra are not computed again: the factor rides on the inverse transform's load.  Filtered
 *       with the pixel window (:644-647), signalNorm = y0 / (spline value at centreYX_d) (:654-659); for N > 1 the map maximum
 *       and the N filtered planes' values at its position go to relWeights_d (N values, then the maximum: fRelWeights of
 *       :661-667 are their float32 ratios); data maps come out in float64.
 *   outputUnits 1 ('uK', :669-685): unit templates only, signalNorm = 1 / max of the filtered template in float32 arithmetic;
 *       data maps come out in float32.
 *   Then :696-756: inverse transform of the data spectra times the filter with the float32 cast, signalNorm and point-source
 *   mask applied at the store, noise map (grid mode), optional edge trim, nf_finalize; and, if axisWinY_d / axisWinX_d are given,
 *   the pixel-window undo as two nf_axis_filter passes (tables for pow = -1 as that entry point wants them).
 * Scales run in groups of up to scaleBatch (consecutive scales whose data_d buffers lie map after map; at least as many groups
 * as lanes): a group's stages are single launches with the scales as batch dimension, per scale the arithmetic of the entry
 * points above.  Groups are dealt round-robin to nLanes lanes; lane 0 runs on `stream`, the others on streams the library keeps
 * per device, forked from and joined back into `stream` with events - so for the caller the whole call is ordinary work on `stream`.
 * work_d: nLanes device buffers of nf_matched_filter_workspace_bytes(tile) bytes each (256-byte aligned).
 * At most nf_matched_filter_max_lanes() lanes are used (4: MI355X boxes give a process 4 hardware queues).  The lane streams
 * and their events are made once per device, by the first nf_matched_filter_workspace_bytes call on it.  Threads: calls on one
 * device may come from several host threads - a call holds the device's lane mutex from its fork to its join, so they take turns -
 * but two calls in flight at once must not share a workspace buffer, and each orders its work by the stream it is given. */
#define NF_MF_MAXFREQ 8                /* room in the per-frequency arrays of the two structs */
#define NF_MF_MAXN 6                   /* most frequency maps a call takes: the limit of the filter build (nf_build_filter*) */
typedef struct nf_mf_tile {
    int H, W, N;                       /* map shape, number of frequency maps (<= NF_MF_MAXN; NF_ERR_SIZE beyond) */
    int prec;                          /* NF_F32 / NF_F64: precision of fMaps_d */
    int covType;                       /* NF_F32 / NF_F64: type of noiseCov_d */
    int outputUnits;                   /* 0 = yc, 1 = uK */
    const void* fMaps_d;               /* [N][H][W] complex, half-stored: spectra of the apodised maps (nf_fft2_forward_half) */
    const void* noiseCov_d;            /* [N][N][H][W] smoothed noise covariance (nf_cross_power_half + nf_gaussian_filter) */
    double w[NF_MF_MAXFREQ];           /* frequency weights (nemo/filters.py:587-608) */
    const double* A_d;                 /* painter geometry, as nf_paint_radial takes it */
    const double* B_d;
    const double* C_d;
    const unsigned char* psMask_d;     /* point-source mask, survey mask (may be NULL = all ones) */
    const unsigned char* surveyMask_d;
    const void* medWeights_d;          /* median weight map for the noise map */
    int wtype;                         /* NF_F32 / NF_F64 of medWeights_d */
    int ny, nx;                        /* noise grid (nf_noise_map) */
    const int* yEdges_d;
    const int* xEdges_d;
    int overlap, numBins, estimator;
    const double* quantiles;           /* numBins HOST doubles */
    int trimSizePix;                   /* edge trim (nf_edge_check size), 0 = none */
    int apodPix;
    double y0;                         /* yc: amplitude of the normalisation template (2e-4 in the reference) */
    const double* centreYX_d;          /* yc: (y, x) of the tile centre, 2 device doubles */
    int centreClass;                   /* yc: pointsClass of that point for nf_spline_eval */
    int centreRowLo, centreRowHi;      /* yc, N == 1: the rows of the filtered template that evaluation reads (0, 0: all) */
    int halfWin;
    const double* winY_d;              /* yc: pixel-window tables for pow = +1 (nf_fft2_inverse_half) */
    const double* winX_d;
    const double* axisWinY_d;          /* pixel-window undo (both or neither): nf_axis_filter tables for axis 1 and axis 0 */
    const double* axisWinX_d;
    unsigned int noiseTablesLanes;     /* bit l: work_d[l] still holds this tile's noise block tables (tablesValid of nf_noise_map
                                        * for the first group of lane l; later groups of a lane in one call always reuse) */
    int scaleBatch;                    /* scales that walk the sequence together, as the batch dimension of every launch (1..8;
                                        * 0 = 1).  The workspace holds that many slots: its size grows in proportion.  Small tiles
                                        * want 8 (the chip is filled by the batch, not by one map); a 16 Mpix tile fills it alone */
} nf_mf_tile;

typedef struct nf_mf_scale {
    /* per map: log-uniform radial table of the unit template (as nf_paint_radial takes it) and - yc - the factor that turns
     * it into the y0-scaled template of the normalisation (templateScale) */
    const double* r_d[NF_MF_MAXFREQ];
    const double* prof_d[NF_MF_MAXFREQ];
    const double* slope_d[NF_MF_MAXFREQ];
    int ntab[NF_MF_MAXFREQ];
    double r0[NF_MF_MAXFREQ], dlnr[NF_MF_MAXFREQ];
    double amp[NF_MF_MAXFREQ], sign[NF_MF_MAXFREQ], templateScale[NF_MF_MAXFREQ];
    /* outputs */
    float* filt_d;                     /* [N][H][W] the filter, or NULL: the cube is not wanted (it is applied all the same) */
    double* signalNorm_d;              /* 1 double (uK: the float32 value, widened) */
    double* relWeights_d;              /* yc, N > 1: N + 1 doubles (see above); else unused, may be NULL */
    void* data_d;                      /* [H][W] filtered map; yc float64, uK float32 */
    void* sn_d;                        /* [H][W] S/N map, type of data_d */
    float* rms_d;                      /* [H][W] noise map */
    unsigned char* mask_d;             /* [H][W] survey mask after edge trim / apodisation border */
    unsigned long long* snChunkRange_d; /* optional (may be NULL): the S/N range table of nf_finalize for sn_d */
} nf_mf_scale;

size_t nf_matched_filter_workspace_bytes(const nf_mf_tile* tile);
int nf_matched_filter_max_lanes(void);
/* The scalar step between the filtered template and the filtered map, on the device so that the host does not wait for it:
 * outputUnits 0: out_d[0] = y0 / src_d[0] in float64 (nemo/filters.py:659, src_d = spline value of the filtered template at the
 * tile centre); 1: out_d[0] = 1 / float32(src_d[0]) in float32 arithmetic, widened (:684, src_d = nf_map_max result). */
int nf_signal_norm(int outputUnits, const double* src_d, double y0, double* out_d, void* stream);
int nf_matched_filter_scales(nf_plan* plan, const nf_mf_tile* tile, const nf_mf_scale* scales, int nScales,
                             void* const* work_d, int nLanes, void* stream);

/* Instrumentation (no counterpart in the reference; used by bench.py): when enabled, every entry point above brackets
 * the work it enqueues with a HIP event pair on the caller's stream.  nf_profile_enable(on) switches recording and
 * clears earlier records; nf_profile_read synchronises the recorded events of one tag and returns the number of calls,
 * their summed device time (ms) and their summed algorithmic HBM bytes (inputs read once + outputs written once).
 * Tags 0..nf_profile_num_tags()-1 are named by nf_profile_tag_name (fft_rows_f64, fft_cols_f64, ..., spline_eval). */
int nf_profile_enable(int on);
/* restrict the recording to the entry points whose tag bit is set (default: all).  An event pair costs a few microseconds of
 * queue time; a throughput run that only needs one entry point's timing records only that one. */
int nf_profile_select(unsigned long long tagMask);
int nf_profile_num_tags(void);
const char* nf_profile_tag_name(int tag);
int nf_profile_read(int tag, long long* count, double* totalMs, double* totalBytes);

#ifdef __cplusplus
}
#endif
#endif

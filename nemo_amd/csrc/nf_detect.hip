// Detection kernels (K15-K17, K9): photometry.getObjectPositions / findObjects (nemo/photometry.py:59-75,
// 81-84,128,213-248).
//
//  label_components  threshold (strict >) + 4-connected component labelling by union-find on pixel indices
//                    (root = smallest raster index of the component), then consecutive numbering of the roots
//                    in raster order - the numbering scipy.ndimage.label produces.
//  object_stats      per label: pixel count, sum v, sum v*y, sum v*x (exact integer accumulation of v * 2^40 with integer
//                    atomics - order-independent, so positions are bit-reproducible; aggregated over runs of equal labels
//                    so a large component does not serialise on a few addresses), maximum value and
//                    the first raster position that attains it (ndimage.maximum_position).
//  ring_flags        nemo/photometry.py:65-75: a component with > minRingPix pixels whose centre of mass does
//                    not land on one of its own pixels is a ring.
//  spline_eval       RectBivariateSpline(kx=ky=3, s=0)(y, x): the interpolating bicubic spline with not-a-knot
//                    ends, evaluated from a +-halfWin pixel window (the influence of samples further away decays
//                    as 0.268^k; at 32 pixels it is below 1e-18).  A window that touches
//                    the true map edge applies the not-a-knot condition there, as FITPACK does.
#include "../../include/nemofilt.h"

#include "nf_common.h"
#include "nf_kernels.h"

#include <mutex>

namespace nf {

static inline int grid_for(long long n, int threads) {
    long long g = (n + threads - 1) / threads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename VT>
__device__ __forceinline__ bool is_fg(const VT* sn, long long p, VT thr, int negate) {
    VT v = sn[p];
    if (negate) v = v * (VT)-1;
    return v > thr;
}

#define NF_SCAN_CHUNK 1024   // pixels per workgroup (256 threads x 4) in every labelling pass

// At a detection threshold of a few sigma almost every 1024-pixel chunk of the map holds no pixel above it.  The first
// pass records that per chunk (chunkFG), and every later pass leaves such a chunk after reading one flag: the union-find
// passes then cost what the foreground costs, and only the first pass (read the map, write parent) and the last (write
// the labels) touch the whole map.
// The maps of one launch (blockIdx.y = map): the scales of a tile are labelled together (nf_label_and_stats_batch), every pass
// ONE launch for all of them.  work[m] = parent[n], blockCounts[nblocks + 1], chunkFG[nblocks].
struct DetBatch {
    const void* sn[NF_SB_MAX];
    int* labels[NF_SB_MAX];
    int* work[NF_SB_MAX];
    int* dirty[NF_SB_MAX];                             // kept labels (k_relabel), or NULL
    double* stats[NF_SB_MAX];                          // the record table (cleared by the first labelling pass when asked)
    const unsigned long long* snRange[NF_SB_MAX];      // nf_finalize's S/N range per chunk, or NULL
    const int* chunkFG[NF_SB_MAX];                     // the sums: foreground flags of the labelling call that made the labels, or NULL
    const int* count[NF_SB_MAX];                       // the sums: where the number of labels found lies on the device, or NULL
};

template <typename VT>
__global__ void __launch_bounds__(256) k_ccl_init(DetBatch b, VT thr, int negate, long long n, long long clearWords) {
    const int m = blockIdx.y, nblocks = gridDim.x;
    const VT* __restrict__ sn = (const VT*)b.sn[m];
    int* __restrict__ parent = b.work[m];
    int* __restrict__ chunkFG = parent + n + nblocks + 1;
    const unsigned long long* __restrict__ snRange = b.snRange[m];
    // (label_and_stats: the record table of the sums that follow is cleared here rather than by a fill of its own)
    unsigned long long* __restrict__ clear = (unsigned long long*)b.stats[m];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < clearWords; i += (long long)gridDim.x * 256) clear[i] = 0ULL;
    if (snRange != nullptr) {
        // the chunk's largest / smallest S/N is known (nf_finalize wrote it): a chunk that cannot hold foreground is not read
        VT top = negate ? (VT)nf_key_value(snRange[gridDim.x + blockIdx.x]) * (VT)-1 : (VT)nf_key_value(snRange[blockIdx.x]);
        if (!(top > thr)) {
            if (threadIdx.x == 0) chunkFG[blockIdx.x] = 0;
            return;
        }
    }
    const long long base = (long long)blockIdx.x * NF_SCAN_CHUNK;
    int any = 0;
    bool fg[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        fg[j] = p < n && is_fg<VT>(sn, p, thr, negate);
        any |= fg[j] ? 1 : 0;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) chunkFG[blockIdx.x] = any;
    if (!any) return;            // nobody reads `parent` inside a chunk without foreground (every pass asks chunkFG first)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        if (p < n) parent[p] = fg[j] ? (int)p : -1;
    }
}

__device__ __forceinline__ int uf_find(const int* parent, int x) {
    int p = parent[x];
    while (p != x) {
        x = p;
        p = parent[x];
    }
    return x;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            int t = a;
            a = b;
            b = t;
        }
        int old = atomicMin(&parent[a], b);   // link the larger root under the smaller index
        if (old == a) return;
        a = old;
    }
}

__global__ void __launch_bounds__(256) k_ccl_merge(DetBatch b, int H, int W) {
    const long long n = (long long)H * W, base = (long long)blockIdx.x * NF_SCAN_CHUNK;
    int* __restrict__ parent = b.work[blockIdx.y];
    const int* __restrict__ chunkFG = parent + n + gridDim.x + 1;
    if (!chunkFG[blockIdx.x]) return;
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        if (p >= n || parent[p] < 0) continue;
        int x = (int)(p % W);
        // (a neighbour in a chunk without foreground is background: its `parent` entry was never written)
        if (x > 0 && chunkFG[(p - 1) / NF_SCAN_CHUNK] && parent[p - 1] >= 0) uf_union(parent, (int)p, (int)(p - 1));
        if (p >= W && chunkFG[(p - W) / NF_SCAN_CHUNK] && parent[p - W] >= 0) uf_union(parent, (int)p, (int)(p - W));
    }
}

// every foreground pixel points at its root; the roots of the chunk are counted on the way
__global__ void __launch_bounds__(256) k_ccl_flatten_count(DetBatch b, long long n) {
    __shared__ int red[4];
    int* __restrict__ parent = b.work[blockIdx.y];
    int* __restrict__ blockCounts = parent + n;
    const int* __restrict__ chunkFG = blockCounts + gridDim.x + 1;
    if (!chunkFG[blockIdx.x]) {
        if (threadIdx.x == 0) blockCounts[blockIdx.x] = 0;
        return;
    }
    const long long base = (long long)blockIdx.x * NF_SCAN_CHUNK;
    int c = 0;
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        if (p < n && parent[p] >= 0) {
            const int r = uf_find(parent, (int)p);
            parent[p] = r;
            c += r == (int)p ? 1 : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blockCounts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// The roots of chunk b are numbered from 1 + (roots in the chunks before b), raster order.  Every chunk with foreground sums the
// counts before it by itself (at most 64 KB of coalesced reads from the L2 - cheaper than a scan kernel between two launches);
// the last workgroup also leaves the total in blockCounts[nblocks].
__global__ void __launch_bounds__(256) k_number_roots(DetBatch b, long long n) {
    __shared__ int waveTot[4], redTot[4];
    const int nblocks = gridDim.x;
    const int* __restrict__ parent = b.work[blockIdx.y];
    int* __restrict__ blockCounts = b.work[blockIdx.y] + n;
    const int* __restrict__ chunkFG = blockCounts + nblocks + 1;
    int* __restrict__ labels = b.labels[blockIdx.y];
    const bool fg = chunkFG[blockIdx.x] != 0, last = (int)blockIdx.x == nblocks - 1;
    if (!fg && !last) return;
    int before = 0;
    {
        int acc = 0;
        const int upto = (int)blockIdx.x;
        for (int i0 = threadIdx.x; i0 < upto; i0 += 256 * 8) {
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * 256 < upto ? blockCounts[i0 + u * 256] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) redTot[threadIdx.x >> 6] = acc;
        __syncthreads();
        before = redTot[0] + redTot[1] + redTot[2] + redTot[3];
        if (last && threadIdx.x == 0) blockCounts[nblocks] = before + blockCounts[blockIdx.x];
    }
    if (!fg) return;
    long long base = (long long)blockIdx.x * NF_SCAN_CHUNK;
    int flags[4], c = 0;
    for (int j = 0; j < 4; ++j) {
        long long p = base + threadIdx.x * 4 + j;
        flags[j] = (p < n && parent[p] == (int)p) ? 1 : 0;
        c += flags[j];
    }
    // exclusive prefix of c over the workgroup: wave scan + wave totals
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) waveTot[wave] = incl;
    __syncthreads();
    int pre = incl - c;
    for (int w = 0; w < wave; ++w) pre += waveTot[w];
    int id = before + pre;
    for (int j = 0; j < 4; ++j) {
        long long p = base + threadIdx.x * 4 + j;
        if (p < n && flags[j]) labels[p] = ++id;   // ids start at 1
    }
}

// `dirty` (optional): the caller keeps `labels` between calls - zero everywhere except in chunks whose dirty flag is set (they
// held foreground in the previous call).  A chunk without foreground is then cleared only if it is dirty, and the flags are
// brought up to date: the pass writes what the foreground touches, not the whole map.
__global__ void __launch_bounds__(256) k_relabel(DetBatch b, long long n) {
    const int* __restrict__ parent = b.work[blockIdx.y];
    const int* __restrict__ chunkFG = parent + n + gridDim.x + 1;
    int* __restrict__ labels = b.labels[blockIdx.y];
    int* __restrict__ dirty = b.dirty[blockIdx.y];
    const long long base = (long long)blockIdx.x * NF_SCAN_CHUNK;
    if (!chunkFG[blockIdx.x]) {
        if (dirty == nullptr || dirty[blockIdx.x]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long p = base + j * 256 + threadIdx.x;
                if (p < n) labels[p] = 0;
            }
        }
        __syncthreads();      // (every wavefront has read the flag before thread 0 rewrites it)
        if (dirty != nullptr && threadIdx.x == 0) dirty[blockIdx.x] = 0;
        return;
    }
    if (dirty != nullptr && threadIdx.x == 0) dirty[blockIdx.x] = 1;
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        if (p >= n) continue;
        int r = parent[p];
        if (r < 0)
            labels[p] = 0;
        else if (r != (int)p)
            labels[p] = labels[r];
    }
}

long long label_chunk_flags_offset(int H, int W) {
    const long long n = (long long)H * W;
    return n + nf_div_up(n, NF_SCAN_CHUNK) + 1;
}

// the five labelling passes for the nb maps of `b` (one launch each); clearWords > 0: b.stats[m] is cleared on the way
static int label_batch(int vtype, const DetBatch& b, int nb, double threshold, int negate, int H, int W, long long clearWords,
                       hipStream_t s, bool relabel = true) {
    const long long n = (long long)H * W;
    if (n >= 2147483647LL) {
        nf_set_error("label_components: map too large for 32-bit pixel indices");
        return NF_ERR_SIZE;
    }
    dim3 g(nf_div_up(n, NF_SCAN_CHUNK), nb), t(256);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_ccl_init<double>, g, t, 0, s, b, threshold, negate, n, clearWords);
    else  // a Python-float threshold compared against a float32 map is cast to float32 by numpy
        hipLaunchKernelGGL(k_ccl_init<float>, g, t, 0, s, b, (float)threshold, negate, n, clearWords);
    hipLaunchKernelGGL(k_ccl_merge, g, t, 0, s, b, H, W);
    hipLaunchKernelGGL(k_ccl_flatten_count, g, t, 0, s, b, n);
    hipLaunchKernelGGL(k_number_roots, g, t, 0, s, b, n);
    if (relabel) hipLaunchKernelGGL(k_relabel, g, t, 0, s, b, n);      // (else: k_object_sums<.., true> does it on its way)
    NF_CHECK_LAUNCH("label_components");
    return NF_OK;
}

int label_components(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work,
                     int* numObjects, int* dirty, hipStream_t s) {
    DetBatch b = {};
    b.sn[0] = sn;
    b.labels[0] = labels;
    b.work[0] = work;
    b.dirty[0] = dirty;
    int rc = label_batch(vtype, b, 1, threshold, negate, H, W, 0, s);
    if (rc != NF_OK) return rc;
    if (numObjects != nullptr) {
        const long long n = (long long)H * W;
        NF_HIP(hipMemcpyAsync(numObjects, work + n + nf_div_up(n, NF_SCAN_CHUNK), sizeof(int), hipMemcpyDeviceToHost, s));
        NF_HIP(hipStreamSynchronize(s));
    }
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// stats layout per label id (0 .. numObjects): [0] count [1] sum v [2] sum v*y [3] sum v*x [4] max v [5] argmax (raster)
#define NF_STAT_STRIDE 6
#define NF_RUN 8

// Sums that do not depend on the order of the additions: every value v is taken as the integer I_v = rint(v * 2^40) (v > 0:
// foreground lies above a threshold >= 0; the quantisation, 9e-13 absolute, is far below the 1e-9 position tolerance), and
// sum I_v, sum I_v * y, sum I_v * x are accumulated EXACTLY in 128-bit integers spread over three 64-bit words each (32-bit
// limbs, so the per-word sums cannot overflow), with integer atomics.  Integer addition is associative: the catalogue
// positions come out bit for bit the same on every run, whatever the scheduling - which float64 atomics cannot promise.
#define NF_FIX_SHIFT 40
#define NF_FIX_VMAX 8.0e6    // values are clamped to [0, 8e6] before the conversion (8e6 * 2^40 < 2^63)
#define NF_LIMB_WORDS 9      // 3 sums x 3 limbs per label

__device__ __forceinline__ void flush_limbs(unsigned long long* w, unsigned __int128 v) {
    const unsigned long long l0 = (unsigned long long)(v & 0xFFFFFFFFu), l1 = (unsigned long long)((v >> 32) & 0xFFFFFFFFu),
                             l2 = (unsigned long long)(v >> 64);
    if (l0) atomicAdd(&w[0], l0);
    if (l1) atomicAdd(&w[1], l1);
    if (l2) atomicAdd(&w[2], l2);
}

__device__ __forceinline__ void flush_run(double* stats, unsigned long long* limbs, int id, double cnt, unsigned __int128 sv,
                                          unsigned __int128 svy, unsigned __int128 svx) {
    unsafeAtomicAdd(&stats[(long long)id * NF_STAT_STRIDE], cnt);      // whole numbers below 2^53: exact in any order
    unsigned long long* w = limbs + (long long)id * NF_LIMB_WORDS;
    flush_limbs(w, sv);
    flush_limbs(w + 3, svy);
    flush_limbs(w + 6, svx);
}

__device__ __forceinline__ unsigned long long ordered_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ULL) ? ~b : (b | 0x8000000000000000ULL);
}

// One workgroup per 1024-pixel chunk (chunkFG, from the labelling pass, when the caller has it: a chunk without foreground
// costs one flag); a thread takes 4 consecutive pixels and flushes one entry per run of equal labels within a row.
// The record table of a map: 6 doubles per label, then the 9 integer limbs, the maximum key and the argmax of every label.
// RELABEL (label_and_stats): the labelling's last pass rides along - a pixel's label is its root's (k_number_roots wrote the
// roots' labels in the launch before; roots are not rewritten here), written to `labels` and summed in the same visit, the
// kept-labels bookkeeping of k_relabel included.  Same labels, same records; one launch less in the chain.
template <typename VT, bool RELABEL>
__global__ void __launch_bounds__(256) k_object_sums(DetBatch b, int negate, int H, int W, int maxId) {
    const int* __restrict__ chunkFG = b.chunkFG[blockIdx.y];
    const long long n = (long long)H * W, p0 = (long long)blockIdx.x * NF_SCAN_CHUNK + threadIdx.x * 4;
    int* __restrict__ labels = b.labels[blockIdx.y];
    if (RELABEL) {
        int* __restrict__ dirty = b.dirty[blockIdx.y];
        if (!chunkFG[blockIdx.x]) {
            if (dirty == nullptr || dirty[blockIdx.x]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p0 + j < n) labels[p0 + j] = 0;
            }
            // (every thread has read the flag before thread 0 rewrites it)
            __syncthreads();
            if (dirty != nullptr && threadIdx.x == 0) dirty[blockIdx.x] = 0;
            return;
        }
        if (dirty != nullptr && threadIdx.x == 0) dirty[blockIdx.x] = 1;
    } else if (chunkFG != nullptr && !chunkFG[blockIdx.x]) {
        return;
    }
    if (p0 >= n) return;
    const VT* __restrict__ sn = (const VT*)b.sn[blockIdx.y];
    const int* __restrict__ parent = b.work[blockIdx.y];
    double* __restrict__ stats = b.stats[blockIdx.y];
    unsigned long long* __restrict__ limbs = (unsigned long long*)(stats + ((size_t)maxId + 1) * NF_STAT_STRIDE);
    unsigned long long* __restrict__ maxKey = limbs + ((size_t)maxId + 1) * NF_LIMB_WORDS;
    int y = (int)(p0 / W), x = (int)(p0 - (long long)y * W);
    int cur = 0;
    double cnt = 0, mx = 0;
    unsigned __int128 sv = 0, svy = 0, svx = 0;
    for (int j = 0; j < 4 && p0 + j < n; ++j) {
        int id;
        if (RELABEL) {
            const int r = parent[p0 + j];
            id = r < 0 ? 0 : labels[r];
            if (r != (int)(p0 + j)) labels[p0 + j] = id;
        } else {
            id = labels[p0 + j];
        }
        if (id > maxId) id = 0;        // (a record table with room for maxId labels: the caller repeats with more room)
        if (id != cur) {
            if (cur > 0) {
                flush_run(stats, limbs, cur, cnt, sv, svy, svx);
                atomicMax(&maxKey[cur], ordered_key(mx));
            }
            cur = id;
            cnt = 0;
            sv = svy = svx = 0;
            mx = -INFINITY;
        }
        if (id > 0) {
            double v = (double)sn[p0 + j];
            if (negate) v = -v;
            const unsigned long long iv = (unsigned long long)rint(fmin(fmax(v, 0.0), NF_FIX_VMAX) * (double)(1ULL << NF_FIX_SHIFT));
            cnt += 1.0;
            sv += iv;
            svy += (unsigned __int128)iv * (unsigned)y;
            svx += (unsigned __int128)iv * (unsigned)x;
            mx = fmax(mx, v);
        }
        if (++x == W) {
            x = 0;
            ++y;
        }
    }
    if (cur > 0) {
        flush_run(stats, limbs, cur, cnt, sv, svy, svx);
        atomicMax(&maxKey[cur], ordered_key(mx));
    }
}

template <typename VT>
__global__ void __launch_bounds__(256) k_object_argmax(DetBatch b, int negate, long long n, int maxId) {
    const int* __restrict__ chunkFG = b.chunkFG[blockIdx.y];
    if (chunkFG != nullptr && !chunkFG[blockIdx.x]) return;
    const VT* __restrict__ sn = (const VT*)b.sn[blockIdx.y];
    const int* __restrict__ labels = b.labels[blockIdx.y];
    const unsigned long long* __restrict__ maxKey =
        (const unsigned long long*)(b.stats[blockIdx.y] + ((size_t)maxId + 1) * NF_STAT_STRIDE) + ((size_t)maxId + 1) * NF_LIMB_WORDS;
    unsigned long long* __restrict__ argmax = (unsigned long long*)maxKey + ((size_t)maxId + 1);
    const long long base = (long long)blockIdx.x * NF_SCAN_CHUNK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long p = base + j * 256 + threadIdx.x;
        if (p >= n) continue;
        int id = labels[p];
        if (id <= 0 || id > maxId) continue;
        double v = (double)sn[p];
        if (negate) v = -v;
        if (ordered_key(v) == maxKey[id]) atomicMax(&argmax[id], ~(unsigned long long)p);    // (inverted: 0 = none, the largest = the first pixel)
    }
}

__device__ __forceinline__ double limbs_to_double(const unsigned long long* w) {
    const unsigned __int128 t = (unsigned __int128)w[0] + ((unsigned __int128)w[1] << 32) + ((unsigned __int128)w[2] << 64);
    const double hi = (double)(unsigned long long)(t >> 64), lo = (double)(unsigned long long)(t & 0xFFFFFFFFFFFFFFFFULL);
    return (hi * 18446744073709551616.0 + lo) * (1.0 / (double)(1ULL << NF_FIX_SHIFT));
}

__global__ void k_stats_finish(DetBatch b, int numObjects) {
    double* __restrict__ stats = b.stats[blockIdx.y];
    const unsigned long long* __restrict__ limbs = (const unsigned long long*)(stats + ((size_t)numObjects + 1) * NF_STAT_STRIDE);
    const unsigned long long* __restrict__ maxKey = limbs + ((size_t)numObjects + 1) * NF_LIMB_WORDS;
    const unsigned long long* __restrict__ argmax = maxKey + ((size_t)numObjects + 1);
    const int* __restrict__ count = b.count[blockIdx.y];
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= numObjects; id += gridDim.x * blockDim.x) {
        const unsigned long long* w = limbs + (long long)id * NF_LIMB_WORDS;
        double* r = stats + (long long)id * NF_STAT_STRIDE;
        r[1] = limbs_to_double(w);
        r[2] = limbs_to_double(w + 3);
        r[3] = limbs_to_double(w + 6);
        unsigned long long k = maxKey[id];
        unsigned long long kb = (k & 0x8000000000000000ULL) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
        r[4] = __longlong_as_double((long long)kb);
        r[5] = (double)~argmax[id];
        // label_and_stats: the number of labels found travels in the (otherwise empty) background record
        if (id == 0 && count != nullptr) r[0] = (double)count[0];
    }
}

// the sums of the nb maps of `b` (one launch per pass); b.stats[m] holds (numObjects+1) * 17 doubles
static int stats_batch(int vtype, const DetBatch& b, int nb, int negate, int H, int W, int numObjects, hipStream_t s, bool relabel = false) {
    const long long n = (long long)H * W;
    dim3 g(nf_div_up(n, NF_SCAN_CHUNK), nb), t(256);
    if (vtype == NF_F64) {
        if (relabel)
            hipLaunchKernelGGL((k_object_sums<double, true>), g, t, 0, s, b, negate, H, W, numObjects);
        else
            hipLaunchKernelGGL((k_object_sums<double, false>), g, t, 0, s, b, negate, H, W, numObjects);
        hipLaunchKernelGGL(k_object_argmax<double>, g, t, 0, s, b, negate, n, numObjects);
    } else {
        if (relabel)
            hipLaunchKernelGGL((k_object_sums<float, true>), g, t, 0, s, b, negate, H, W, numObjects);
        else
            hipLaunchKernelGGL((k_object_sums<float, false>), g, t, 0, s, b, negate, H, W, numObjects);
        hipLaunchKernelGGL(k_object_argmax<float>, g, t, 0, s, b, negate, n, numObjects);
    }
    hipLaunchKernelGGL(k_stats_finish, dim3(grid_for((long long)numObjects + 1, 256), nb), dim3(256), 0, s, b, numObjects);
    NF_CHECK_LAUNCH("object_stats");
    return NF_OK;
}

// `stats` must hold (numObjects+1) * 17 doubles: 6 per record, then scratch - the 9 integer limbs, the maximum key and the
// argmax of every label
int object_stats(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* stats,
                 const int* chunkFG, hipStream_t s) {
    const size_t nrec = (size_t)numObjects + 1;
    // records, limbs, maximum keys and (inverted) argmax all start at zero
    NF_HIP(hipMemsetAsync(stats, 0, nrec * (NF_STAT_STRIDE + NF_LIMB_WORDS + 2) * sizeof(double), s));
    DetBatch b = {};
    b.sn[0] = sn;
    b.labels[0] = (int*)labels;
    b.stats[0] = stats;
    b.chunkFG[0] = chunkFG;
    return stats_batch(vtype, b, 1, negate, H, W, numObjects, s);
}

// Labelling and the per-label sums without a host round trip in between, for up to NF_SB_MAX maps of one shape per launch:
// every record table has room for maxObjects labels, the number found stays on the device and comes back in stats[m][0] (the
// background record's count, otherwise unused).
int label_and_stats_batch(int vtype, int nb, const void* const* sn, double threshold, int negate, int* const* labels, int H, int W,
                          int* const* work, int* const* dirty, int maxObjects, double* const* stats,
                          const unsigned long long* const* snRange, hipStream_t s) {
    const long long clearWords = ((long long)maxObjects + 1) * (NF_STAT_STRIDE + NF_LIMB_WORDS + 2);
    const long long n = (long long)H * W;
    const int nblocks = nf_div_up(n, NF_SCAN_CHUNK);
    for (int m0 = 0; m0 < nb; m0 += NF_SB_MAX) {
        const int c = nb - m0 < NF_SB_MAX ? nb - m0 : NF_SB_MAX;
        DetBatch b = {};
        for (int i = 0; i < c; ++i) {
            b.sn[i] = sn[m0 + i];
            b.labels[i] = labels[m0 + i];
            b.work[i] = work[m0 + i];
            b.dirty[i] = dirty != nullptr ? dirty[m0 + i] : nullptr;
            b.stats[i] = stats[m0 + i];
            b.snRange[i] = snRange != nullptr ? snRange[m0 + i] : nullptr;
            b.count[i] = work[m0 + i] + n + nblocks;
            b.chunkFG[i] = work[m0 + i] + n + nblocks + 1;
        }
        int rc = label_batch(vtype, b, c, threshold, negate, H, W, clearWords, s, false);
        if (rc != NF_OK) return rc;
        rc = stats_batch(vtype, b, c, negate, H, W, maxObjects, s, true);
        if (rc != NF_OK) return rc;
    }
    return NF_OK;
}

int label_and_stats(int vtype, const void* sn, double threshold, int negate, int* labels, int H, int W, int* work, int* dirty,
                    int maxObjects, double* stats, const unsigned long long* snRange, hipStream_t s) {
    return label_and_stats_batch(vtype, 1, &sn, threshold, negate, &labels, H, W, &work, &dirty, maxObjects, &stats, &snRange, s);
}

// ---------------------------------------------------------------------------------------------------
// Segment moments for the SExtractor-style shapes of photometry.findObjects (measureShapes, nemo/photometry.py:133-184):
// per label the bounding box and sum v, sum v*dx, sum v*dy, sum v*dx^2, sum v*dy^2, sum v*dx*dy with dx = x - xMin,
// dy = y - yMin (the reference shifts the coordinates by the box corner before it forms the moments, :143-150).
// Same exact integer accumulation as the centroid sums above (I_v * dx^2 < 2^63 * 2^30: inside 128 bits).
// shapes layout per label id: [0] xMin [1] yMin [2] xMax [3] yMax [4] S [5] Sx [6] Sy [7] Sxx [8] Syy [9] Sxy
#define NF_SHAPE_STRIDE 10
#define NF_SHAPE_LIMBS 18

__global__ void k_shape_bbox(const int* __restrict__ labels, int H, int W, int* __restrict__ mins, int* __restrict__ maxs) {
    const int segsPerRow = (W + NF_RUN - 1) / NF_RUN;
    const long long nseg = (long long)H * segsPerRow;
    for (long long sidx = blockIdx.x * (long long)blockDim.x + threadIdx.x; sidx < nseg; sidx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(sidx / segsPerRow), x0 = (int)(sidx % segsPerRow) * NF_RUN;
        int cur = 0, xa = 0, xb = 0;
        for (int j = 0; j < NF_RUN && x0 + j < W; ++j) {
            const int id = labels[(long long)y * W + x0 + j];
            if (id != cur) {
                if (cur > 0) {
                    atomicMin(&mins[2 * cur], xa);
                    atomicMin(&mins[2 * cur + 1], y);
                    atomicMax(&maxs[2 * cur], xb);
                    atomicMax(&maxs[2 * cur + 1], y);
                }
                cur = id;
                xa = x0 + j;
            }
            xb = x0 + j;
        }
        if (cur > 0) {
            atomicMin(&mins[2 * cur], xa);
            atomicMin(&mins[2 * cur + 1], y);
            atomicMax(&maxs[2 * cur], xb);
            atomicMax(&maxs[2 * cur + 1], y);
        }
    }
}

template <typename VT>
__global__ void k_shape_moments(const VT* __restrict__ sn, int negate, const int* __restrict__ labels, int H, int W,
                                const int* __restrict__ mins, unsigned long long* __restrict__ limbs) {
    const int segsPerRow = (W + NF_RUN - 1) / NF_RUN;
    const long long nseg = (long long)H * segsPerRow;
    for (long long sidx = blockIdx.x * (long long)blockDim.x + threadIdx.x; sidx < nseg; sidx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(sidx / segsPerRow), x0 = (int)(sidx % segsPerRow) * NF_RUN;
        int cur = 0, xMin = 0, yMin = 0;
        unsigned __int128 acc[6] = {0, 0, 0, 0, 0, 0};
        for (int j = 0; j < NF_RUN && x0 + j < W; ++j) {
            const long long p = (long long)y * W + x0 + j;
            const int id = labels[p];
            if (id != cur) {
                if (cur > 0) {
                    unsigned long long* w = limbs + (long long)cur * NF_SHAPE_LIMBS;
                    for (int k = 0; k < 6; ++k) flush_limbs(w + 3 * k, acc[k]);
                }
                cur = id;
                for (int k = 0; k < 6; ++k) acc[k] = 0;
                if (id > 0) {
                    xMin = mins[2 * id];
                    yMin = mins[2 * id + 1];
                }
            }
            if (id > 0) {
                double v = (double)sn[p];
                if (negate) v = -v;
                const unsigned long long iv = (unsigned long long)rint(fmin(fmax(v, 0.0), NF_FIX_VMAX) * (double)(1ULL << NF_FIX_SHIFT));
                const unsigned long long dx = (unsigned long long)(x0 + j - xMin), dy = (unsigned long long)(y - yMin);
                acc[0] += iv;
                acc[1] += (unsigned __int128)iv * dx;
                acc[2] += (unsigned __int128)iv * dy;
                acc[3] += (unsigned __int128)iv * (dx * dx);
                acc[4] += (unsigned __int128)iv * (dy * dy);
                acc[5] += (unsigned __int128)iv * (dx * dy);
            }
        }
        if (cur > 0) {
            unsigned long long* w = limbs + (long long)cur * NF_SHAPE_LIMBS;
            for (int k = 0; k < 6; ++k) flush_limbs(w + 3 * k, acc[k]);
        }
    }
}

__global__ void k_shape_finish(double* __restrict__ shapes, const unsigned long long* __restrict__ limbs, const int* __restrict__ mins,
                               const int* __restrict__ maxs, int numObjects) {
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= numObjects; id += gridDim.x * blockDim.x) {
        double* r = shapes + (long long)id * NF_SHAPE_STRIDE;
        r[0] = (double)mins[2 * id];
        r[1] = (double)mins[2 * id + 1];
        r[2] = (double)maxs[2 * id];
        r[3] = (double)maxs[2 * id + 1];
        const unsigned long long* w = limbs + (long long)id * NF_SHAPE_LIMBS;
        for (int k = 0; k < 6; ++k) r[4 + k] = limbs_to_double(w + 3 * k);
    }
}

// `shapes` must hold (numObjects+1) * 30 doubles: 10 per record, then scratch - 18 integer limbs and the box corners
int object_shapes(int vtype, const void* sn, int negate, const int* labels, int H, int W, int numObjects, double* shapes, hipStream_t s) {
    const size_t nrec = (size_t)numObjects + 1;
    unsigned long long* limbs = (unsigned long long*)(shapes + nrec * NF_SHAPE_STRIDE);
    int* mins = (int*)(limbs + nrec * NF_SHAPE_LIMBS);
    int* maxs = mins + 2 * nrec;
    NF_HIP(hipMemsetAsync(limbs, 0, nrec * NF_SHAPE_LIMBS * sizeof(unsigned long long), s));
    NF_HIP(hipMemsetAsync(mins, 0x7F, 2 * nrec * sizeof(int), s));
    NF_HIP(hipMemsetAsync(maxs, 0, 2 * nrec * sizeof(int), s));
    const long long nseg = (long long)H * ((W + NF_RUN - 1) / NF_RUN);
    dim3 g(grid_for(nseg, 256)), b(256);
    hipLaunchKernelGGL(k_shape_bbox, g, b, 0, s, labels, H, W, mins, maxs);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_shape_moments<double>, g, b, 0, s, (const double*)sn, negate, labels, H, W, mins, limbs);
    else
        hipLaunchKernelGGL(k_shape_moments<float>, g, b, 0, s, (const float*)sn, negate, labels, H, W, mins, limbs);
    hipLaunchKernelGGL(k_shape_finish, dim3(grid_for(nrec, 256)), dim3(256), 0, s, shapes, limbs, mins, maxs, numObjects);
    NF_CHECK_LAUNCH("object_shapes");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// completeness.getRMSTab (nemo/completeness.py:1329-1333): the distinct non-zero values of the noise map and the survey area
// at each.  The noise map is piecewise constant (one value per noise cell and weight bin), so the table is small: an
// open-addressing hash table keyed by the value's bits, filled with integer atomics - pixel counts, and areas as exact
// multiples of 2^-50 deg^2 (a float32 pixel area above 2^-27 deg^2 is such a multiple: sums do not depend on the order).
// A thread takes NF_RUN consecutive pixels of a row and inserts one entry per run of equal values.
#define NF_RMS_AREA_SHIFT 50

__device__ __forceinline__ void rms_insert(unsigned* __restrict__ keys, unsigned long long* __restrict__ cnt,
                                           unsigned long long* __restrict__ area, unsigned mask, unsigned key, unsigned long long n,
                                           unsigned long long a, int* __restrict__ overflow) {
    unsigned h = (key * 2654435761u) >> 7;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        h &= mask;
        const unsigned old = atomicCAS(&keys[h], 0u, key);
        if (old == 0u || old == key) {
            atomicAdd(&cnt[h], n);
            if (a) atomicAdd(&area[h], a);
            return;
        }
        ++h;
    }
    atomicOr(overflow, 1);
}

__global__ void k_rms_table(const float* __restrict__ rms, const unsigned char* __restrict__ areaMask, const int* __restrict__ flagMask,
                            int maxFlags, const unsigned char* __restrict__ intersect, const unsigned long long* __restrict__ rowAreaFix,
                            int H, int W, unsigned* __restrict__ keys, unsigned long long* __restrict__ cnt,
                            unsigned long long* __restrict__ area, unsigned mask, int* __restrict__ overflow) {
    const int segsPerRow = (W + NF_RUN - 1) / NF_RUN;
    const long long nseg = (long long)H * segsPerRow;
    for (long long sidx = blockIdx.x * (long long)blockDim.x + threadIdx.x; sidx < nseg; sidx += (long long)gridDim.x * blockDim.x) {
        const int y = (int)(sidx / segsPerRow), x0 = (int)(sidx % segsPerRow) * NF_RUN;
        const unsigned long long ra = rowAreaFix[y];
        unsigned cur = 0;
        unsigned long long n = 0, a = 0;
        for (int j = 0; j < NF_RUN && x0 + j < W; ++j) {
            const long long p = (long long)y * W + x0 + j;
            float v = rms[p];
            bool in = areaMask == nullptr || areaMask[p] != 0;
            if (flagMask != nullptr && flagMask[p] > maxFlags) in = false;
            if (intersect != nullptr && intersect[p] == 0) {
                in = false;
                v = 0.0f;
            }
            const unsigned key = v == 0.0f ? 0u : __float_as_uint(v);
            if (key != cur) {
                if (cur != 0u) rms_insert(keys, cnt, area, mask, cur, n, a, overflow);
                cur = key;
                n = 0;
                a = 0;
            }
            if (key != 0u) {
                ++n;
                if (in) a += ra * (unsigned long long)(areaMask ? areaMask[p] : 1);
            }
        }
        if (cur != 0u) rms_insert(keys, cnt, area, mask, cur, n, a, overflow);
    }
}

// keys: tableSize (a power of two) uint32, zeroed by this call; cnt / area: tableSize uint64 each
int rms_table(const float* rms, const unsigned char* areaMask, const int* flagMask, int maxFlags, const unsigned char* intersect,
              const unsigned long long* rowAreaFix, int H, int W, unsigned* keys, unsigned long long* cnt, unsigned long long* area,
              int tableSize, int* overflow, hipStream_t s) {
    if (tableSize < 64 || (tableSize & (tableSize - 1)) != 0) {
        nf_set_error("rms_table: the table size must be a power of two >= 64");
        return NF_ERR_ARG;
    }
    NF_HIP(hipMemsetAsync(keys, 0, (size_t)tableSize * sizeof(unsigned), s));
    NF_HIP(hipMemsetAsync(cnt, 0, (size_t)tableSize * sizeof(unsigned long long), s));
    NF_HIP(hipMemsetAsync(area, 0, (size_t)tableSize * sizeof(unsigned long long), s));
    NF_HIP(hipMemsetAsync(overflow, 0, sizeof(int), s));
    const long long nseg = (long long)H * ((W + NF_RUN - 1) / NF_RUN);
    hipLaunchKernelGGL(k_rms_table, dim3(grid_for(nseg, 256)), dim3(256), 0, s, rms, areaMask, flagMask, maxFlags, intersect, rowAreaFix, H, W,
                       keys, cnt, area, (unsigned)(tableSize - 1), overflow);
    NF_CHECK_LAUNCH("k_rms_table");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// np.argmax / ndarray.max of a whole map (nemo/filters.py:661,684): maximum value and FIRST raster index attaining it.
// One pass: every thread keeps (value, first index) of its grid-stride share, the workgroup reduces with wavefront
// shuffles + LDS and writes ONE partial; a second single-workgroup launch reduces the partials.  No atomics (65536
// wavefronts hammering one 8-byte word made the first version atomics-bound: 218 us for a 67 MB map).
#define NF_MAXRED_BLOCKS 2048
#define NF_MAXRED_THREADS 256

__device__ __forceinline__ void max_first(double& v, long long& i, double ov, long long oi) {
    // maximum value; among equals the smallest index (np.argmax returns the first occurrence)
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ void block_max_first(double& best, long long& arg) {
    __shared__ double sv[NF_MAXRED_THREADS / 64];
    __shared__ long long si[NF_MAXRED_THREADS / 64];
    for (int off = 32; off > 0; off >>= 1) {
        double ov = __shfl_down(best, off, 64);
        long long oi = __shfl_down(arg, off, 64);
        max_first(best, arg, ov, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sv[wave] = best;
        si[wave] = arg;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NF_MAXRED_THREADS / 64; ++w) max_first(best, arg, sv[w], si[w]);
    }
}

// blockIdx.y: one map per entry of a batch - maps lie entry after entry (mapStride elements apart), each entry has its own
// NF_MAP_MAX_DOUBLES block of results and partials
template <typename VT>
__global__ void __launch_bounds__(NF_MAXRED_THREADS) k_map_max(const VT* __restrict__ m0, long long n, long long mapStride,
                                                                double* __restrict__ mapMax0) {
    const VT* __restrict__ m = m0 + (long long)blockIdx.y * mapStride;
    double* __restrict__ pv = mapMax0 + (long long)blockIdx.y * NF_MAP_MAX_DOUBLES + 4;
    long long* __restrict__ pi = (long long*)(pv + NF_MAXRED_BLOCKS);
    double best = -INFINITY;
    long long arg = 0x7FFFFFFFFFFFFFFFLL;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const double v = (double)m[p];
        if (v > best) {      // indices grow along the loop: strict > keeps the first occurrence
            best = v;
            arg = p;
        }
    }
    block_max_first(best, arg);
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = best;
        pi[blockIdx.x] = arg;
    }
}
// blockIdx.x: batch entry
__global__ void __launch_bounds__(NF_MAXRED_THREADS) k_map_max_finish(double* __restrict__ mapMax0, int nPart) {
    double* __restrict__ out = mapMax0 + (long long)blockIdx.x * NF_MAP_MAX_DOUBLES;
    const double* __restrict__ pv = out + 4;
    const long long* __restrict__ pi = (const long long*)(pv + NF_MAXRED_BLOCKS);
    double best = -INFINITY;
    long long arg = 0x7FFFFFFFFFFFFFFFLL;
    for (int k = threadIdx.x; k < nPart; k += blockDim.x) max_first(best, arg, pv[k], pi[k]);
    block_max_first(best, arg);
    if (threadIdx.x == 0) {
        out[0] = best;
        out[1] = (double)arg;
    }
}

// nf_sum_planes and the first half of nf_map_max in one pass over the planes (the per-scale call with several frequencies:
// nemo/filters.py:847 then :661): out = in[0] + ... + in[N-1] in plane order, and per workgroup the maximum of out with its
// first index.  blockIdx.y: batch entry ([entry][N][n] planes in, [entry][n] sums out).
__global__ void __launch_bounds__(NF_MAXRED_THREADS) k_sum_planes_max(const float* __restrict__ in0, float* __restrict__ out0, int N,
                                                                       long long n, double* __restrict__ mapMax0) {
    const float* __restrict__ in = in0 + (long long)blockIdx.y * N * n;
    float* __restrict__ out = out0 + (long long)blockIdx.y * n;
    double* __restrict__ pv = mapMax0 + (long long)blockIdx.y * NF_MAP_MAX_DOUBLES + 4;
    long long* __restrict__ pi = (long long*)(pv + NF_MAXRED_BLOCKS);
    double best = -INFINITY;
    long long arg = 0x7FFFFFFFFFFFFFFFLL;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        float acc = in[p];
        for (int i = 1; i < N; ++i) acc = __fadd_rn(acc, in[(long long)i * n + p]);
        out[p] = acc;
        const double v = (double)acc;
        if (v > best) {
            best = v;
            arg = p;
        }
    }
    block_max_first(best, arg);
    if (threadIdx.x == 0) {
        pv[blockIdx.x] = best;
        pi[blockIdx.x] = arg;
    }
}

// the second half of nf_map_max, the relative-weight gather (nemo/filters.py:661-667: the N planes at the position of the
// maximum, then the maximum) and signalNorm = y0 / peak (:659) as one finishing kernel.  blockIdx.x: batch entry.
__global__ void __launch_bounds__(NF_MAXRED_THREADS) k_map_max_finish_weights(double* __restrict__ mapMax0, int nPart,
                                                                               const float* __restrict__ planes0, int N, long long npix,
                                                                               nf::BatchPtrs relOutP, const double* __restrict__ peak0,
                                                                               double y0, nf::BatchPtrs normOutP,
                                                                               double* __restrict__ normWs) {
    double* __restrict__ out = mapMax0 + (long long)blockIdx.x * NF_MAP_MAX_DOUBLES;
    const double* __restrict__ pv = out + 4;
    const long long* __restrict__ pi = (const long long*)(pv + NF_MAXRED_BLOCKS);
    const float* __restrict__ planes = planes0 + (long long)blockIdx.x * N * npix;
    double* __restrict__ relOut = (double*)relOutP.p[blockIdx.x];
    double* __restrict__ normOut = (double*)normOutP.p[blockIdx.x];
    double best = -INFINITY;
    long long arg = 0x7FFFFFFFFFFFFFFFLL;
    for (int k = threadIdx.x; k < nPart; k += blockDim.x) max_first(best, arg, pv[k], pi[k]);
    block_max_first(best, arg);
    __shared__ long long at;
    if (threadIdx.x == 0) {
        out[0] = best;
        out[1] = (double)arg;
        at = arg;
        relOut[N] = best;
        const double norm = __ddiv_rn(y0, peak0[blockIdx.x]);
        normOut[0] = norm;
        if (normWs != nullptr) normWs[blockIdx.x] = norm;      // (the batch's norms side by side, for the next launch)
    }
    __syncthreads();
    if ((int)threadIdx.x < N) relOut[threadIdx.x] = (double)planes[(long long)threadIdx.x * npix + at];
}

static int map_max_parts(long long n) {
    long long want = (n + NF_MAXRED_THREADS * 8LL - 1) / (NF_MAXRED_THREADS * 8LL);
    return (int)(want < 1 ? 1 : (want > NF_MAXRED_BLOCKS ? NF_MAXRED_BLOCKS : want));
}

// (batched forms: nb maps / plane stacks one after the other, mapMax = nb blocks of NF_MAP_MAX_DOUBLES, peak = nb doubles)
int sum_planes_max(const float* in, float* out, int N, long long n, double* mapMax, hipStream_t s, int nb) {
    hipLaunchKernelGGL(k_sum_planes_max, dim3(map_max_parts(n), nb), dim3(NF_MAXRED_THREADS), 0, s, in, out, N, n, mapMax);
    NF_CHECK_LAUNCH("k_sum_planes_max");
    return NF_OK;
}

int map_max_finish_weights_batch(double* mapMax, long long n, const float* planes, int N, const BatchPtrs& relOut, const double* peak,
                                 double y0, const BatchPtrs& normOut, double* normWs, int nb, hipStream_t s) {
    hipLaunchKernelGGL(k_map_max_finish_weights, dim3(nb), dim3(NF_MAXRED_THREADS), 0, s, mapMax, map_max_parts(n), planes, N, n, relOut,
                       peak, y0, normOut, normWs);
    NF_CHECK_LAUNCH("k_map_max_finish_weights");
    return NF_OK;
}

int map_max_finish_weights(double* mapMax, long long n, const float* planes, int N, double* relOut, const double* peak, double y0,
                           double* normOut, hipStream_t s) {
    BatchPtrs r, o;
    memset(&r, 0, sizeof(r));
    memset(&o, 0, sizeof(o));
    r.p[0] = relOut;
    o.p[0] = normOut;
    return map_max_finish_weights_batch(mapMax, n, planes, N, r, peak, y0, o, nullptr, 1, s);
}

// out: NF_MAP_MAX_DOUBLES (= 4 + 2 * 2048) doubles on the device, owned by the caller: [0] max value, [1] first index of
// the max, [4..] per-workgroup partials (values, then indices); nb maps (mapStride elements apart): nb such blocks
int map_max(int vtype, const void* m, long long n, double* out, hipStream_t s, int nb, long long mapStride) {
    const int nPart = map_max_parts(n);
    dim3 g(nPart, nb), b(NF_MAXRED_THREADS);
    if (vtype == NF_F64)
        hipLaunchKernelGGL(k_map_max<double>, g, b, 0, s, (const double*)m, n, mapStride, out);
    else
        hipLaunchKernelGGL(k_map_max<float>, g, b, 0, s, (const float*)m, n, mapStride, out);
    hipLaunchKernelGGL(k_map_max_finish, dim3(nb), b, 0, s, out, nPart);
    NF_CHECK_LAUNCH("map_max");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
__global__ void k_ring_flags(const int* __restrict__ ringLabels, const double* __restrict__ ringStats, int numRing, int minRingPix,
                             int H, int W, unsigned char* __restrict__ isRing) {
    for (int id = blockIdx.x * blockDim.x + threadIdx.x; id <= numRing; id += gridDim.x * blockDim.x) {
        unsigned char flag = 0;
        if (id > 0) {
            const double* r = ringStats + (long long)id * NF_STAT_STRIDE;
            if (r[0] > (double)minRingPix) {
                double y = r[2] / r[1], x = r[3] / r[1];
                int iy = (int)y, ix = (int)x;   // Python int(): truncation
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    if (ringLabels[(long long)iy * W + ix] != id) flag = 1;
                }
            }
        }
        isRing[id] = flag;
    }
}

int ring_flags(const int* ringLabels, const double* ringStats, int numRing, int minRingPix, int H, int W, unsigned char* isRing,
               hipStream_t s) {
    hipLaunchKernelGGL(k_ring_flags, dim3(grid_for(numRing + 1, 256)), dim3(256), 0, s, ringLabels, ringStats, numRing, minRingPix,
                       H, W, isRing);
    NF_CHECK_LAUNCH("k_ring_flags");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
#define NF_SPL_MAXN 72

// Points further than halfWin + 2 pixels from every map edge: the second derivatives of the interpolating cubic spline
// on a uniform grid are M_i = sum_j G(i - j) d_j with d_j = 6 (y_{j-1} - 2 y_j + y_{j+1}) and the Green's function of the
// (1, 4, 1) system, G(k) = rho^|k| / (2 sqrt 3), rho = sqrt 3 - 2: the boundary conditions are forgotten like
// |rho|^distance (0.268^33 ~ 1e-19), which is why a +-32 pixel window reproduces the whole-map FITPACK fit
// (SURVEY.md Appendix A).  The value at (py, px) is then a separable weighted sum over the window - no sequential
// tridiagonal solves, one wavefront-parallel dot product per point.
__device__ __forceinline__ bool spline_interior(int iy, int ix, int H, int W, int halfWin) {
    return iy - halfWin - 1 >= 0 && iy + halfWin + 2 <= H - 1 && ix - halfWin - 1 >= 0 && ix + halfWin + 2 <= W - 1;
}

__device__ __forceinline__ double ipow(double b, int k) {
    double r = 1.0;
    while (k > 0) {
        if (k & 1) r *= b;
        b *= b;
        k >>= 1;
    }
    return r;
}

// weights w[m], m = 0 .. 2K+3, of samples i-K-1 .. i+K+2 for an evaluation at i + t  (c[] = scratch of 2K+4 doubles)
__device__ __forceinline__ void spline_weights(double t, int K, double* c, double* w, int lane, int nlanes) {
    const double rho = -0.26794919243112270647255365849413;    // sqrt(3) - 2
    const double A = 0.28867513459481288225457439025098;       // 1 / (2 sqrt(3))
    const double t1 = 1.0 - t;
    const double a = t1 * t1 * t1 / 6.0 - t1 / 6.0, b = t * t * t / 6.0 - t / 6.0;
    const int n = 2 * K + 4;
    // c_j for j = i-K-1 .. i+K+2 (index m = j - (i-K-1)); zero on the two outermost samples
    for (int m = lane; m < n; m += nlanes) {
        double v = 0.0;
        if (m >= 1 && m <= n - 2) {
            const int k0 = abs(m - (K + 1));       // |i - j|
            const int k1 = abs(m - (K + 2));       // |i + 1 - j|
            v = A * (a * ipow(rho, k0) + b * ipow(rho, k1));
        }
        c[m] = v;
    }
    __syncthreads();
    for (int m = lane; m < n; m += nlanes) {
        const double cm = c[m], cl = m > 0 ? c[m - 1] : 0.0, cr = m < n - 1 ? c[m + 1] : 0.0;
        double v = 6.0 * (cl - 2.0 * cm + cr);
        if (m == K + 1) v += t1;
        if (m == K + 2) v += t;
        w[m] = v;
    }
    __syncthreads();
}

template <typename VT>
__device__ __forceinline__ void spline_eval_interior(const VT* __restrict__ map, int negate, int W, double py, double px, int iy,
                                                     int ix, int K, double* lds /* 3 * NF_SPL_MAXN + 2 doubles */,
                                                     double* __restrict__ outp) {
    double *cbuf = lds, *wy = lds + NF_SPL_MAXN, *wx = lds + 2 * NF_SPL_MAXN, *red = lds + 3 * NF_SPL_MAXN;
    const int n = 2 * K + 4;
    spline_weights(py - (double)iy, K, cbuf, wy, threadIdx.x, blockDim.x);
    spline_weights(px - (double)ix, K, cbuf, wx, threadIdx.x, blockDim.x);
    const int y0 = iy - K - 1, x0 = ix - K - 1;
    double acc = 0.0;
    if ((int)threadIdx.x < n) {
        const VT* col = map + (long long)y0 * W + x0 + threadIdx.x;
        // 16 independent row loads in flight per round (the loop is latency-bound otherwise: one HBM round trip per row)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = 0;
        for (; r + 15 < n; r += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (double)col[(long long)(r + j) * W];
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                s0 += wy[r + j] * v[j];
                s1 += wy[r + j + 1] * v[j + 1];
                s2 += wy[r + j + 2] * v[j + 2];
                s3 += wy[r + j + 3] * v[j + 3];
            }
        }
        for (; r < n; ++r) s0 += wy[r] * (double)col[(long long)r * W];
        acc = ((s0 + s1) + (s2 + s3)) * wx[threadIdx.x];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = red[0] + red[1];
        *outp = negate ? -v : v;
    }
}

// ---------------------------------------------------------------------------------------------------
// 1-D not-a-knot cubic spline through n >= 4 unit-spaced samples y[0..n-1], evaluated at position u in [0, n-1].
// Second derivatives M: M_1 = d_1/6 and M_{n-2} = d_{n-2}/6 follow directly from the not-a-knot conditions on a uniform
// grid (d_i = 6 (y_{i-1} - 2 y_i + y_{i+1})); between them the constant (1, 4, 1) system M_{i-1} + 4 M_i + M_{i+1} = d_i,
// i = 2 .. n-3, is a Dirichlet problem.  Instead of a sequential elimination (two dependent sweeps of n steps) it is
// solved in closed form: the particular solution P_i = sum_j G(i - j) d_j with the Green's function of the infinite
// system (G(k) = rho^|k| / (2 sqrt 3), rho = sqrt 3 - 2) plus the two homogeneous solutions rho^(i-1), rho^(n-2-i) fitted
// to the two known end values.  Only the two M the evaluation interval needs are formed: four independent dot products
// of length n - 4, no dependent chain.  rw[k] = rho^k, k < NF_SPL_MAXN.
__device__ double nak_spline_eval(const double* y, int n, double u, const double* rw) {
    const double A = 0.28867513459481288225457439025098;       // 1 / (2 sqrt(3))
    int i = (int)floor(u);
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    // the two interior second derivatives to form: those of the interval, or the end pair an extrapolated end value needs
    const int qa = i == 0 ? 1 : (i + 1 == n - 1 ? n - 3 : i), qb = qa + 1;
    double p1 = 0.0, pn = 0.0, pa = 0.0, pb = 0.0;
    double ym = y[1], yc = y[2];
    for (int j = 2; j <= n - 3; ++j) {
        const double yp = y[j + 1];
        const double dj = 6.0 * (ym - 2.0 * yc + yp);
        p1 += rw[j - 1] * dj;
        pn += rw[n - 2 - j] * dj;
        pa += rw[abs(qa - j)] * dj;
        pb += rw[abs(qb - j)] * dj;
        ym = yc;
        yc = yp;
    }
    p1 *= A;
    pn *= A;
    pa *= A;
    pb *= A;
    const double r1 = (y[0] - 2.0 * y[1] + y[2]) - p1, rn = (y[n - 3] - 2.0 * y[n - 2] + y[n - 1]) - pn;   // d/6 - P
    const double e = rw[n - 3], det = 1.0 - e * e;
    const double alpha = (r1 - e * rn) / det, beta = (rn - e * r1) / det;
    const double Ma = pa + alpha * rw[qa - 1] + beta * rw[n - 2 - qa];
    const double Mb = pb + alpha * rw[qb - 1] + beta * rw[n - 2 - qb];
    double Mi, Mi1;
    if (i == 0) {                       // M_0 = 2 M_1 - M_2
        Mi = 2.0 * Ma - Mb;
        Mi1 = Ma;
    } else if (i + 1 == n - 1) {        // M_{n-1} = 2 M_{n-2} - M_{n-3}
        Mi = Mb;
        Mi1 = 2.0 * Mb - Ma;
    } else {
        Mi = Ma;
        Mi1 = Mb;
    }
    const double t = u - (double)i, t1 = 1.0 - t;
    const double y0 = y[i], y1 = y[i + 1];
    return Mi * t1 * t1 * t1 / 6.0 + Mi1 * t * t * t / 6.0 + (y0 - Mi / 6.0) * t1 + (y1 - Mi1 / 6.0) * t;
}

// interior points: small static LDS, so many workgroups share a CU (the windowed-solve kernel below needs 85 KB each)
template <typename VT>
__global__ void __launch_bounds__(128) k_spline_eval_interior(const VT* __restrict__ map, int negate, int H, int W,
                                                              const double* __restrict__ yx, int halfWin, double* __restrict__ out) {
    __shared__ double lds[3 * NF_SPL_MAXN + 2];
    const double py = yx[2 * blockIdx.x], px = yx[2 * blockIdx.x + 1];
    if (!(py >= 0.0 && py <= (double)(H - 1) && px >= 0.0 && px <= (double)(W - 1))) return;
    const int iy = (int)floor(py), ix = (int)floor(px);
    if (!spline_interior(iy, ix, H, W, halfWin)) return;
    spline_eval_interior<VT>(map, negate, W, py, px, iy, ix, halfWin, lds, out + blockIdx.x);
}

// one point by the windowed solve (points whose window meets the map's border); shared memory as k_spline_eval lays it out
template <typename VT>
__device__ __forceinline__ void spline_eval_windowed(const VT* __restrict__ map, int negate, int H, int W, double py, double px,
                                                     int halfWin, unsigned char* smem, double* __restrict__ outp) {
    double* win = (double*)smem;                            // [ny][nxp]
    int iy = (int)floor(py), ix = (int)floor(px);
    int y0 = max(0, iy - halfWin), y1 = min(H - 1, iy + 1 + halfWin);
    int x0 = max(0, ix - halfWin), x1 = min(W - 1, ix + 1 + halfWin);
    const int ny = y1 - y0 + 1, nx = x1 - x0 + 1;
    const int nxp = nx + 1;
    double* colv = win + NF_SPL_MAXN * (NF_SPL_MAXN + 1);    // [ny]
    double* rw = colv + NF_SPL_MAXN;                         // [NF_SPL_MAXN] rho^k
    if (!(py >= 0.0 && py <= (double)(H - 1) && px >= 0.0 && px <= (double)(W - 1)) || ny < 4 || nx < 4) {
        if (threadIdx.x == 0) *outp = nan("");
        return;
    }
    if (spline_interior(iy, ix, H, W, halfWin)) return;      // the interior kernel wrote this point
#pragma unroll 8
    for (int i = threadIdx.x; i < ny * nx; i += blockDim.x) {
        int r = i / nx, c = i % nx;
        double v = (double)map[(long long)(y0 + r) * W + x0 + c];
        win[r * nxp + c] = negate ? -v : v;
    }
    if (threadIdx.x < NF_SPL_MAXN) rw[threadIdx.x] = ipow(-0.26794919243112270647255365849413, (int)threadIdx.x);
    __syncthreads();
    for (int r = threadIdx.x; r < ny; r += blockDim.x) colv[r] = nak_spline_eval(win + r * nxp, nx, px - (double)x0, rw);
    __syncthreads();
    if (threadIdx.x == 0) *outp = nak_spline_eval(colv, ny, py - (double)y0, rw);
}

template <typename VT>
__global__ void k_spline_eval(const VT* __restrict__ map, int negate, int H, int W, const double* __restrict__ yx, int halfWin,
                              const int* __restrict__ sel, double* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    const int pt = sel ? sel[blockIdx.x] : (int)blockIdx.x; // `sel`: the points this kernel serves (the others are interior)
    spline_eval_windowed<VT>(map, negate, H, W, yx[2 * pt], yx[2 * pt + 1], halfWin, nf_smem, out + pt);
}

// Several (map, point list) jobs in one launch: blockIdx.y is the job.  What a detection pass reads off its maps at the object
// positions - S/N, the filtered map, the reference scale's two maps - is one launch of each kernel instead of one per map.
template <typename VT>
__global__ void __launch_bounds__(128) k_spline_batch_interior(SplineBatch b, int H, int W, int halfWin) {
    __shared__ double lds[3 * NF_SPL_MAXN + 2];
    const int j = blockIdx.y, pt = blockIdx.x;
    if (pt >= b.n[j]) return;
    const double py = b.yx[j][2 * pt], px = b.yx[j][2 * pt + 1];
    if (!(py >= 0.0 && py <= (double)(H - 1) && px >= 0.0 && px <= (double)(W - 1))) return;
    const int iy = (int)floor(py), ix = (int)floor(px);
    if (!spline_interior(iy, ix, H, W, halfWin)) return;
    spline_eval_interior<VT>((const VT*)b.map[j], b.negate[j], W, py, px, iy, ix, halfWin, lds, b.out[j] + pt);
}

template <typename VT>
__global__ void k_spline_batch_windowed(SplineBatch b, int H, int W, int halfWin) {
    extern __shared__ __align__(16) unsigned char nf_smem[];
    const int j = blockIdx.y;
    if ((int)blockIdx.x >= b.nEdge[j]) return;
    const int pt = b.edge[j] ? b.edge[j][blockIdx.x] : (int)blockIdx.x;
    spline_eval_windowed<VT>((const VT*)b.map[j], b.negate[j], H, W, b.yx[j][2 * pt], b.yx[j][2 * pt + 1], halfWin, nf_smem, b.out[j] + pt);
}

int spline_eval(int vtype, const void* map, int negate, int H, int W, const double* yx, int nPts, int halfWin, int pointsClass,
                const int* edgeSel, int nEdge, double* out, hipStream_t s) {
    if (nPts <= 0) return NF_OK;
    if (2 * halfWin + 2 > NF_SPL_MAXN) {
        nf_set_error("spline_eval: window too large");
        return NF_ERR_ARG;
    }
    size_t lds = (size_t)(NF_SPL_MAXN * (NF_SPL_MAXN + 1) + 2 * NF_SPL_MAXN) * sizeof(double);
    static bool attr[2] = {false, false};
    if (vtype == NF_F64) {
        if (!attr[1]) {
            NF_HIP(hipFuncSetAttribute((const void*)k_spline_eval<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[1] = true;
        }
        if (pointsClass != 2)
            hipLaunchKernelGGL(k_spline_eval_interior<double>, dim3(nPts), dim3(128), 0, s, (const double*)map, negate, H, W, yx, halfWin, out);
        if (pointsClass != 1)
            hipLaunchKernelGGL(k_spline_eval<double>, dim3(edgeSel ? nEdge : nPts), dim3(128), lds, s, (const double*)map, negate, H, W, yx,
                               halfWin, edgeSel, out);
    } else {
        if (!attr[0]) {
            NF_HIP(hipFuncSetAttribute((const void*)k_spline_eval<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[0] = true;
        }
        if (pointsClass != 2)
            hipLaunchKernelGGL(k_spline_eval_interior<float>, dim3(nPts), dim3(128), 0, s, (const float*)map, negate, H, W, yx, halfWin, out);
        if (pointsClass != 1)
            hipLaunchKernelGGL(k_spline_eval<float>, dim3(edgeSel ? nEdge : nPts), dim3(128), lds, s, (const float*)map, negate, H, W, yx,
                               halfWin, edgeSel, out);
    }
    NF_CHECK_LAUNCH("k_spline_eval");
    return NF_OK;
}

int spline_eval_batch(int vtype, const SplineBatch& b, int nJobs, int H, int W, int halfWin, hipStream_t s) {
    if (nJobs <= 0) return NF_OK;
    if (2 * halfWin + 2 > NF_SPL_MAXN) {
        nf_set_error("spline_eval_batch: window too large");
        return NF_ERR_ARG;
    }
    int maxPts = 0, maxEdge = 0;
    bool anyInterior = false;
    for (int j = 0; j < nJobs; ++j) {
        maxPts = b.n[j] > maxPts ? b.n[j] : maxPts;
        maxEdge = b.nEdge[j] > maxEdge ? b.nEdge[j] : maxEdge;
        if (b.edge[j] == nullptr || b.nEdge[j] < b.n[j]) anyInterior = true;
    }
    if (maxPts <= 0) return NF_OK;
    const size_t lds = (size_t)(NF_SPL_MAXN * (NF_SPL_MAXN + 1) + 2 * NF_SPL_MAXN) * sizeof(double);
    static bool attr[2] = {false, false};
    const int a = vtype == NF_F64 ? 1 : 0;
    if (!attr[a]) {
        if (a)
            NF_HIP(hipFuncSetAttribute((const void*)k_spline_batch_windowed<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else
            NF_HIP(hipFuncSetAttribute((const void*)k_spline_batch_windowed<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[a] = true;
    }
    if (anyInterior) {
        if (a)
            hipLaunchKernelGGL(k_spline_batch_interior<double>, dim3(maxPts, nJobs), dim3(128), 0, s, b, H, W, halfWin);
        else
            hipLaunchKernelGGL(k_spline_batch_interior<float>, dim3(maxPts, nJobs), dim3(128), 0, s, b, H, W, halfWin);
    }
    if (maxEdge > 0) {
        if (a)
            hipLaunchKernelGGL(k_spline_batch_windowed<double>, dim3(maxEdge, nJobs), dim3(128), lds, s, b, H, W, halfWin);
        else
            hipLaunchKernelGGL(k_spline_batch_windowed<float>, dim3(maxEdge, nJobs), dim3(128), lds, s, b, H, W, halfWin);
    }
    NF_CHECK_LAUNCH("k_spline_batch");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Cross match of a catalogue with itself (catalogs._makeOptimalCatalogForcedRef, nemo/catalogs.py:319-362): rows are unit
// vectors; two rows are partners when their chord is shorter than the match radius' chord (every row is its own partner).
//   k_xmatch_best   per row i: number of partners, and the first partner (lowest index) with the highest SNR - what
//                   xIndices[np.argmax(SNR[xIndices])] picks; a NaN among the partners' SNRs makes numpy's segment maximum
//                   NaN, and then the first partner is taken.
//   k_xmatch_drop   row j is dropped if some partner i has several partners and prefers another row to j.
// A chord within 1e-9 (relative) of the limit is not decided here: *nearFlag is raised and the host repeats the match with
// the exact great-circle formula (catalogs._pairsWithin).
#define NF_XM_THREADS 64

__device__ __forceinline__ int xm_relation(double ax, double ay, double az, double bx, double by, double bz, double lo2, double hi2) {
    const double dz = az - bz;
    if (dz * dz > hi2) return 0;          // most rows differ by more than the radius in declination alone
    const double dx = ax - bx, dy = ay - by;
    const double c2 = dx * dx + dy * dy + dz * dz;
    return c2 < lo2 ? 1 : (c2 <= hi2 ? 2 : 0);
}

#define NF_XM_SLICES 16      // the partner loop of every row is cut into this many slices (blockIdx.y): 16 x more wavefronts

struct XmPart {          // partners of one row inside one slice
    int cnt, first, best, sawNaN;
    double mx;
};

__global__ void __launch_bounds__(NF_XM_THREADS) k_xmatch_best_part(const double* __restrict__ u, const double* __restrict__ snr, int n,
                                                                     double lo2, double hi2, XmPart* __restrict__ part,
                                                                     int* __restrict__ nearFlag) {
    __shared__ double sx[NF_XM_THREADS], sy[NF_XM_THREADS], sz[NF_XM_THREADS], ss[NF_XM_THREADS];
    const int i = blockIdx.x * NF_XM_THREADS + threadIdx.x;
    const bool live = i < n;
    const double ax = live ? u[3 * i] : 0.0, ay = live ? u[3 * i + 1] : 0.0, az = live ? u[3 * i + 2] : 2.0;
    const int per = ((n + NF_XM_SLICES - 1) / NF_XM_SLICES + NF_XM_THREADS - 1) / NF_XM_THREADS * NF_XM_THREADS;
    const int jBeg = blockIdx.y * per, jEnd = min(n, jBeg + per);
    int cnt = 0, bst = -1, first = -1;
    double mx = 0.0;
    bool sawNaN = false, near = false;
    for (int j0 = jBeg; j0 < jEnd; j0 += NF_XM_THREADS) {
        const int jl = j0 + threadIdx.x;
        __syncthreads();
        sx[threadIdx.x] = jl < jEnd ? u[3 * jl] : 0.0;
        sy[threadIdx.x] = jl < jEnd ? u[3 * jl + 1] : 0.0;
        sz[threadIdx.x] = jl < jEnd ? u[3 * jl + 2] : -2.0;
        ss[threadIdx.x] = jl < jEnd ? snr[jl] : 0.0;
        __syncthreads();
        const int m = min(NF_XM_THREADS, jEnd - j0);
        for (int k = 0; k < m; ++k) {
            const int rel = xm_relation(ax, ay, az, sx[k], sy[k], sz[k], lo2, hi2);
            if (rel == 2) near = true;
            if (rel == 1) {
                const double sj = ss[k];
                ++cnt;
                if (first < 0) {
                    first = j0 + k;
                    bst = first;
                    mx = sj;
                    sawNaN = sj != sj;
                } else if (sj != sj) {
                    sawNaN = true;
                } else if (sj > mx) {
                    mx = sj;
                    bst = j0 + k;
                }
            }
        }
    }
    if (live) {
        XmPart p;
        p.cnt = cnt;
        p.first = first;
        p.best = bst;
        p.sawNaN = sawNaN ? 1 : 0;
        p.mx = mx;
        part[(long long)blockIdx.y * n + i] = p;
        if (near) atomicOr(nearFlag, 1);
    }
}

// slices in ascending partner order: the first slice with partners gives `first`; a later slice only takes `best` over with
// a strictly higher SNR (np.argmax keeps the first maximum)
__global__ void k_xmatch_merge(const XmPart* __restrict__ part, int n, int* __restrict__ count, int* __restrict__ best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cnt = 0, first = -1, bst = -1;
    double mx = 0.0;
    bool sawNaN = false;
    for (int sl = 0; sl < NF_XM_SLICES; ++sl) {
        const XmPart p = part[(long long)sl * n + i];
        if (p.cnt == 0) continue;
        cnt += p.cnt;
        sawNaN = sawNaN || p.sawNaN != 0;
        if (first < 0) {
            first = p.first;
            bst = p.best;
            mx = p.mx;
        } else if (p.mx > mx) {
            mx = p.mx;
            bst = p.best;
        }
    }
    count[i] = cnt;
    best[i] = sawNaN ? first : bst;
}

__global__ void __launch_bounds__(NF_XM_THREADS) k_xmatch_drop(const double* __restrict__ u, int n, double lo2, double hi2,
                                                                const int* __restrict__ count, const int* __restrict__ best,
                                                                unsigned char* __restrict__ drop) {
    __shared__ double sx[NF_XM_THREADS], sy[NF_XM_THREADS], sz[NF_XM_THREADS];
    __shared__ int sc[NF_XM_THREADS], sb[NF_XM_THREADS];
    const int j = blockIdx.x * NF_XM_THREADS + threadIdx.x;
    const bool live = j < n;
    const double ax = live ? u[3 * j] : 0.0, ay = live ? u[3 * j + 1] : 0.0, az = live ? u[3 * j + 2] : 2.0;
    const int per = ((n + NF_XM_SLICES - 1) / NF_XM_SLICES + NF_XM_THREADS - 1) / NF_XM_THREADS * NF_XM_THREADS;
    const int iBeg = blockIdx.y * per, iEnd = min(n, iBeg + per);
    bool d = false;
    for (int i0 = iBeg; i0 < iEnd; i0 += NF_XM_THREADS) {
        const int il = i0 + threadIdx.x;
        __syncthreads();
        sx[threadIdx.x] = il < iEnd ? u[3 * il] : 0.0;
        sy[threadIdx.x] = il < iEnd ? u[3 * il + 1] : 0.0;
        sz[threadIdx.x] = il < iEnd ? u[3 * il + 2] : -2.0;
        sc[threadIdx.x] = il < iEnd ? count[il] : 0;
        sb[threadIdx.x] = il < iEnd ? best[il] : -1;
        __syncthreads();
        const int m = min(NF_XM_THREADS, iEnd - i0);
        for (int k = 0; k < m; ++k)
            if (sc[k] > 1 && sb[k] != j && xm_relation(ax, ay, az, sx[k], sy[k], sz[k], lo2, hi2) == 1) d = true;
    }
    if (live && d) drop[j] = 1;        // (every slice that finds a reason writes the same 1; cleared beforehand)
}

size_t xmatch_workspace_bytes(int n) { return (size_t)NF_XM_SLICES * (size_t)n * sizeof(XmPart); }

int xmatch_self(const double* u, const double* snr, int n, double lo2, double hi2, int* count, int* best, unsigned char* drop,
                int* nearFlag, void* work, size_t workBytes, hipStream_t s) {
    if (workBytes < xmatch_workspace_bytes(n)) {
        nf_set_error("xmatch_self: workspace too small");
        return NF_ERR_SIZE;
    }
    NF_HIP(hipMemsetAsync(nearFlag, 0, sizeof(int), s));
    NF_HIP(hipMemsetAsync(drop, 0, (size_t)n, s));
    const int g = nf_div_up(n, NF_XM_THREADS);
    hipLaunchKernelGGL(k_xmatch_best_part, dim3(g, NF_XM_SLICES), dim3(NF_XM_THREADS), 0, s, u, snr, n, lo2, hi2, (XmPart*)work, nearFlag);
    hipLaunchKernelGGL(k_xmatch_merge, dim3(nf_div_up(n, 256)), dim3(256), 0, s, (const XmPart*)work, n, count, best);
    hipLaunchKernelGGL(k_xmatch_drop, dim3(g, NF_XM_SLICES), dim3(NF_XM_THREADS), 0, s, u, n, lo2, hi2, count, best, drop);
    NF_CHECK_LAUNCH("k_xmatch");
    return NF_OK;
}

// ---------------------------------------------------------------------------------------------------
// xmatch_self for callers that cannot allocate (HIP graph capture, serving loops): the per-slice partial results live in a
// file-scope device array instead of a caller workspace.  Each GPU the library runs on holds its own copy of the array (at
// its own address) until the process ends; host code never uses the array's NAME as an address - it asks the runtime for the
// address once per device (hipGetSymbolAddress, below) and passes that pointer to the kernels, as xmatch_self passes `work`.
// Nothing in the array has to start at zero: k_xmatch_best_part writes every record (all NF_XM_SLICES slices of every row
// i < n) before k_xmatch_merge reads it.  All calls on one GPU share the array: calls that overlap on one GPU race.
static __device__ __align__(256) unsigned char nf_xm_scratch[(size_t)NF_XM_SLICES * NF_XM_STATIC_MAX_ROWS * sizeof(XmPart)];

namespace {
const int kMaxDevices = 64;
void* g_xmScratchPtr[kMaxDevices];       // device address of nf_xm_scratch per HIP device, 0 until first asked for
std::mutex g_xmScratchMu;
}  // namespace

int xmatch_static_scratch(void** ptr) {
    int dev = -1;
    NF_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) {
        nf_set_error("xmatch_self_static: unexpected device index");
        return NF_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_xmScratchMu);
    if (!g_xmScratchPtr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(nf_xm_scratch));     // the address on the CURRENT device
        if (e != hipSuccess) return nf_hip_fail(e, "hipGetSymbolAddress(nf_xm_scratch)");
        if (!p) {
            nf_set_error("xmatch_self_static: no device address for the scratch array");
            return NF_ERR_HIP;
        }
        g_xmScratchPtr[dev] = p;
    }
    *ptr = g_xmScratchPtr[dev];
    return NF_OK;
}

int xmatch_self_static(const double* u, const double* snr, int n, double lo2, double hi2, int* count, int* best, unsigned char* drop,
                       int* nearFlag, hipStream_t s) {
    if (n > NF_XM_STATIC_MAX_ROWS) {
        nf_set_error("xmatch_self_static: more than NF_XMATCH_STATIC_MAX_ROWS (16384) rows - use nf_xmatch_self with a workspace");
        return NF_ERR_SIZE;
    }
    void* work = nullptr;
    int rc = xmatch_static_scratch(&work);
    if (rc) return rc;
    return xmatch_self(u, snr, n, lo2, hi2, count, best, drop, nearFlag, work, xmatch_workspace_bytes(n), s);
}

}  // namespace nf

// Integer segment plans: stable CSR build, segment pointers, floor levels, chunk tables.
// All results are bit-exact against oracle/plan.py (numpy stable argsort / searchsorted).
//
// CSR build = counting sort: histogram -> scan -> placement -> per-row ordering.  Keys that are
// already non-decreasing (stories, graphs, cross edges by voxel: the reference's batches are built
// that way) are detected on the device and take the identity permutation without any sorting;
// otherwise slots are claimed with atomics and every row is then put back into COO order
// (insertion sort for short rows, a work list of long rows sorted by whole workgroups).
#include "common.h"

namespace {

// flags layout per build: [0] keys out of range, [1] slots with perm != identity,
//                         [2] order violations (key[e] < key[e-1]), [3] long-row work list length
constexpr int kSmallRow = 16;
constexpr int kLdsRow = 4096;

__device__ __forceinline__ void d_hist(const int64_t* __restrict__ key, int64_t E, int64_t N,
                                       int32_t* __restrict__ count, int32_t* __restrict__ flags) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t base = (int64_t)blockIdx.x * blockDim.x;
    for (; base < E; base += stride) {            // workgroup-uniform trip count
        int64_t e = base + threadIdx.x;
        const bool live = e < E;
        int64_t k = live ? key[e] : -1;
        const bool ok = live && k >= 0 && k < N;
        if (live && !ok) atomicAdd(&flags[0], 1);
        if (live && e > 0 && key[e - 1] > k) atomicAdd(&flags[2], 1);
        // wave-aggregated when the whole wave holds one key (sorted / clustered inputs)
        const int64_t k0 = __shfl(k, 0, 64);
        if (__all(ok && k == k0)) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&count[k0 + 1], 64);
        } else if (ok) {
            atomicAdd(&count[k + 1], 1);
        }
    }
}

__global__ void k_hist(const int64_t* __restrict__ key, int64_t E, int64_t N,
                       int32_t* __restrict__ count, int32_t* __restrict__ flags) {
    d_hist(key, E, N, count, flags);
}

// In-place inclusive scan of a[0..n) by ONE workgroup of 1024 threads (copy != NULL: a second copy of the result).
// Tiles of 4,096 consecutive elements, four per thread: neighbouring lanes read neighbouring words (the first form gave
// every thread ONE contiguous range of n / 1024 elements -- 4 KB apart between lanes at n = 10^6: 3.0 ms per scan of a
// 1,048,576-voxel batch's row counts).  Integer sums: the result does not depend on the order of the additions.
__device__ __forceinline__ void d_scan_inplace(int32_t* __restrict__ a, int64_t n, int32_t* __restrict__ copy) {
    __shared__ int32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int32_t carry = 0;
    for (int64_t base = 0; base < n; base += 4096) {               // uniform trip count
        const int64_t i0 = base + 4 * t;
        int32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? a[i0 + k] : 0;
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        int32_t x = v[3];                                          // inclusive scan of the threads' sums over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t before = 0, tile = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int32_t sw = wsum[w];
            if (w < wave) before += sw;
            tile += sw;
        }
        const int32_t excl = carry + before + (x - v[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) {
                const int32_t r = excl + v[k];
                a[i0 + k] = r;
                if (copy) copy[i0 + k] = r;
            }
        }
        carry += tile;
        __syncthreads();                                           // wsum is written again by the next tile
    }
}

__global__ __launch_bounds__(1024) void k_scan_inplace(int32_t* __restrict__ a, int64_t n) { d_scan_inplace(a, n, nullptr); }

__device__ __forceinline__ void d_place(const int64_t* __restrict__ key, int64_t E, int64_t N,
                                        int32_t* __restrict__ cursor, int32_t* __restrict__ perm,
                                        const int32_t* __restrict__ flags) {
    const bool sorted = flags[2] == 0 && flags[0] == 0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
        if (sorted) { perm[e] = (int32_t)e; continue; }      // stable sort of sorted keys = identity
        int64_t k = key[e];
        if (k < 0 || k >= N) continue;
        int32_t pos = atomicAdd(&cursor[k], 1);
        perm[pos] = (int32_t)e;
    }
}

__global__ void k_place(const int64_t* __restrict__ key, int64_t E, int64_t N,
                        int32_t* __restrict__ cursor, int32_t* __restrict__ perm,
                        const int32_t* __restrict__ flags) {
    d_place(key, E, N, cursor, perm, flags);
}

// all-ascending bitonic network on p[0..n): valid for any n (partners beyond n are +inf)
__device__ void block_bitonic(int32_t* p, int n) {
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int l = i ^ (k - 1);
            if (l > i && l < n) {
                int32_t a = p[i], b = p[l];
                if (a > b) { p[i] = b; p[l] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int l = i ^ j;
                if (l > i && l < n) {
                    int32_t a = p[i], b = p[l];
                    if (a > b) { p[i] = b; p[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// one thread per row: short rows are sorted here, long rows go to the work list (worklist[0..))
__device__ __forceinline__ void d_rowsort_short(const int32_t* __restrict__ rowptr, int64_t N, int32_t* __restrict__ perm,
                                                int32_t* __restrict__ flags, int32_t* __restrict__ worklist) {
    if (flags[2] == 0 && flags[0] == 0) return;
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    const int d = e - b;
    if (d <= 1) return;
    if (d <= kSmallRow) {
        for (int i = b + 1; i < e; ++i) {
            int32_t v = perm[i];
            int j = i - 1;
            while (j >= b && perm[j] > v) { perm[j + 1] = perm[j]; --j; }
            perm[j + 1] = v;
        }
    } else {
        worklist[atomicAdd(&flags[3], 1)] = (int32_t)r;
    }
}

__global__ void k_rowsort_short(const int32_t* __restrict__ rowptr, int64_t N, int32_t* __restrict__ perm,
                                int32_t* __restrict__ flags, int32_t* __restrict__ worklist) {
    d_rowsort_short(rowptr, N, perm, flags, worklist);
}

// workgroups walk the work list; one row at a time per workgroup (LDS when it fits)
__device__ __forceinline__ void d_rowsort_long(const int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                                               const int32_t* __restrict__ flags, const int32_t* __restrict__ worklist) {
    __shared__ int32_t buf[kLdsRow];
    const int n = flags[3];
    for (int w = blockIdx.x; w < n; w += gridDim.x) {      // workgroup-uniform
        const int32_t r = worklist[w];
        const int32_t b = rowptr[r], e = rowptr[r + 1];
        const int d = e - b;
        if (d <= kLdsRow) {
            for (int i = threadIdx.x; i < d; i += blockDim.x) buf[i] = perm[b + i];
            __syncthreads();
            block_bitonic(buf, d);
            for (int i = threadIdx.x; i < d; i += blockDim.x) perm[b + i] = buf[i];
            __syncthreads();
        } else {
            block_bitonic(perm + b, d);
        }
    }
}

__global__ __launch_bounds__(256) void k_rowsort_long(const int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ flags,
                                                      const int32_t* __restrict__ worklist) {
    d_rowsort_long(rowptr, perm, flags, worklist);
}

// Slots [0, rowptr[N]) hold the edges whose key is in range; with out-of-range keys (reported in flags[0]; the caller
// rejects the plan) the slots behind them were never placed: they get defined values instead of being read as indices.
__device__ __forceinline__ void d_fill_col(const int64_t* __restrict__ other, int32_t* __restrict__ perm, int64_t E,
                                           int32_t* __restrict__ col, int32_t* __restrict__ flags,
                                           const int32_t* __restrict__ rowptr, int64_t N) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t valid = rowptr[N];
    int moved = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < E; k += stride) {
        if (k >= valid) { perm[k] = 0; col[k] = 0; continue; }
        int32_t p = perm[k];
        if ((uint32_t)p >= (uint64_t)E) { col[k] = 0; continue; }          // (cannot happen for placed slots)
        col[k] = other ? (int32_t)other[p] : p;
        moved += (p != (int32_t)k);
    }
    if (moved) atomicAdd(&flags[1], moved);
}

__global__ void k_fill_col(const int64_t* __restrict__ other, int32_t* __restrict__ perm, int64_t E,
                           int32_t* __restrict__ col, int32_t* __restrict__ flags, const int32_t* __restrict__ rowptr,
                           int64_t N) {
    d_fill_col(other, perm, E, col, flags, rowptr, N);
}

// ---- the same six steps for up to kMaxBuilds CSR builds at once: blockIdx.y = the build
constexpr int kMaxBuilds = 11;
struct CsrJob { const int64_t* key; const int64_t* other; int64_t E, N; int32_t *rowptr, *col, *perm, *cursor, *flags; };
struct CsrJobs { CsrJob j[kMaxBuilds]; };

__global__ void k_hist_b(CsrJobs J, int32_t* __restrict__ stats) {
    if (stats && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 32)
        stats[threadIdx.x] = (threadIdx.x >= 1 && threadIdx.x <= 4) ? 0x7fffffff : 0;       // 1..4 are minima
    const CsrJob& j = J.j[blockIdx.y];
    d_hist(j.key, j.E, j.N, j.rowptr, j.flags);
}
__global__ __launch_bounds__(1024) void k_scan_b(CsrJobs J) {
    const CsrJob& j = J.j[blockIdx.x];
    d_scan_inplace(j.rowptr, j.N + 1, j.cursor);
}
__global__ void k_place_b(CsrJobs J) {
    const CsrJob& j = J.j[blockIdx.y];
    d_place(j.key, j.E, j.N, j.cursor, j.perm, j.flags);
}
__global__ void k_rowsort_short_b(CsrJobs J) {
    const CsrJob& j = J.j[blockIdx.y];
    d_rowsort_short(j.rowptr, j.N, j.perm, j.flags, j.cursor);
}
__global__ __launch_bounds__(256) void k_rowsort_long_b(CsrJobs J) {
    const CsrJob& j = J.j[blockIdx.y];
    d_rowsort_long(j.rowptr, j.perm, j.flags, j.cursor);
}
__global__ void k_fill_col_b(CsrJobs J) {
    const CsrJob& j = J.j[blockIdx.y];
    d_fill_col(j.other, j.perm, j.E, j.col, j.flags, j.rowptr, j.N);
}

__global__ void k_segptr(const int64_t* __restrict__ seg, int64_t n, int64_t nseg,
                         int32_t* __restrict__ ptr, int32_t* __restrict__ flags) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t cur = seg[i];
    int64_t prev = (i > 0) ? seg[i - 1] : -1;
    if (cur < prev || cur < 0 || cur >= nseg) { atomicAdd(&flags[0], 1); return; }
    for (int64_t s = prev + 1; s <= cur; ++s) ptr[s] = (int32_t)i;
    if (i == n - 1) for (int64_t s = cur + 1; s <= nseg; ++s) ptr[s] = (int32_t)n;
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ void k_floor_min(const int64_t* __restrict__ vfc, const int64_t* __restrict__ vbi, int64_t n, int64_t B,
                            int32_t* __restrict__ first) {
    int64_t base = (int64_t)blockIdx.x * blockDim.x;
    int64_t i = base + threadIdx.x;
    const bool live = i < n;
    int64_t g = live ? vbi[i] : -1;
    const bool ok = live && g >= 0 && g < B;
    int32_t v = ok ? (int32_t)vfc[i] : 0x7fffffff;
    const int64_t g0 = __shfl(g, 0, 64);
    if (__all(ok && g == g0)) {                    // one graph per wave: one atomic per wave
        v = wave_min_i32(v);
        if ((threadIdx.x & 63) == 0) atomicMin(&first[g0], v);
    } else if (ok) {
        atomicMin(&first[g], v);
    }
}

__global__ void k_floor_level(const int64_t* __restrict__ vfc, const int64_t* __restrict__ vbi, int64_t n, int64_t B,
                              const int32_t* __restrict__ first, int32_t* __restrict__ level) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t g = vbi[i];
    level[i] = (g >= 0 && g < B) ? (int32_t)vfc[i] - first[g] : 0;
}

__global__ void k_counts(const int32_t* __restrict__ rowptr, int64_t n, float* __restrict__ count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int d = rowptr[i + 1] - rowptr[i];
    count[i] = (float)(d > 1 ? d : 1);
}

__global__ void k_chunk_counts(const int32_t* __restrict__ rowptr, int64_t n, int32_t R,
                               int32_t* __restrict__ chunk_ptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) chunk_ptr[0] = 0;
    if (i >= n) return;
    int d = rowptr[i + 1] - rowptr[i];
    chunk_ptr[i + 1] = (d + R - 1) / R;
}


// ---------------------------------------------------------------- whole-plan build (bg_plan_build): small helpers
// graph of every story = graph of (any) member voxel (avg_pool_x's batch[perm]); all members agree.  Also fills the
// per-graph first-floor table with +inf for the floor-level pass that follows.
__global__ void k_story_graph(const int64_t* __restrict__ vfc, const int64_t* __restrict__ vbi, int64_t n, int64_t S,
                              int64_t* __restrict__ sg64, int32_t* __restrict__ sg32, int32_t* __restrict__ first, int64_t B) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) first[i] = 0x7fffffff;
    if (i >= n) return;
    const int64_t s = vfc[i];
    if (s < 0 || s >= S) return;
    if (i > 0 && vfc[i - 1] == s) return;              // one writer per run of a story's voxels
    sg64[s] = vbi[i];
    sg32[s] = (int32_t)vbi[i];
}

// Every elementwise table and statistic that depends only on the finished CSRs, as ONE launch: blockIdx.y = the job.
enum { MJ_I64_I32, MJ_ARANGE, MJ_HAS, MJ_COUNTS, MJ_ROWLEN, MJ_LEAVING, MJ_BLKNNZ, MJ_CHUNKCNT, MJ_MAXI32 };
struct MiscJob { const void* a; const void* b; const void* c; void* out; int64_t n; int32_t kind, p0, p1, pad; };
constexpr int kMaxMisc = 32;
struct MiscJobs { MiscJob j[kMaxMisc]; };

__global__ void k_plan_misc(MiscJobs J) {
    const MiscJob& j = J.j[blockIdx.y];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((int64_t)blockIdx.x * blockDim.x >= j.n + (j.kind == MJ_CHUNKCNT ? 1 : 0)) return;      // workgroup-uniform
    const bool live = i < j.n;
    const int32_t* rp = static_cast<const int32_t*>(j.a);
    switch (j.kind) {
        case MJ_I64_I32: if (live) static_cast<int32_t*>(j.out)[i] = (int32_t) static_cast<const int64_t*>(j.a)[i]; break;
        case MJ_ARANGE: if (live) static_cast<int32_t*>(j.out)[i] = (int32_t)i; break;
        case MJ_HAS: if (live) static_cast<float*>(j.out)[i] = rp[i + 1] > rp[i] ? 1.f : 0.f; break;
        case MJ_COUNTS: if (live) { const int d = rp[i + 1] - rp[i]; static_cast<float*>(j.out)[i] = (float)(d > 1 ? d : 1); } break;
        case MJ_CHUNKCNT: {
            int32_t* cp = static_cast<int32_t*>(j.out);
            if (i == 0) cp[0] = 0;
            if (live) cp[i + 1] = (rp[i + 1] - rp[i] + j.p0 - 1) / j.p0;
        } break;
        case MJ_ROWLEN: {                              // min / max row length into stats[p0] / stats[p1] (-1: skip)
            int d_min = 0x7fffffff, d_max = 0;
            if (live) { const int d = rp[i + 1] - rp[i]; d_min = d; d_max = d; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                d_min = min(d_min, __shfl_xor(d_min, off, 64));
                d_max = max(d_max, __shfl_xor(d_max, off, 64));
            }
            int32_t* stats = static_cast<int32_t*>(j.out);
            if ((threadIdx.x & 63) == 0) {
                // (a minimum only falls and a maximum only rises, so a wave whose value cannot improve on the one it
                //  reads -- possibly stale, hence a weaker bound -- has nothing to contribute: at 10^6 rows nearly every
                //  wave, and 16 k atomics on one word are serialised)
                const volatile int32_t* seen = stats;
                if (j.p0 >= 0 && d_min != 0x7fffffff && d_min < seen[j.p0]) atomicMin(&stats[j.p0], d_min);
                if (j.p1 >= 0 && d_max > seen[j.p1]) atomicMax(&stats[j.p1], d_max);
            }
        } break;
        case MJ_MAXI32: {
            int m = live ? rp[i] : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(static_cast<int32_t*>(j.out), m);
        } break;
        case MJ_LEAVING: {                             // edges whose ends lie in different graphs (p0 = number of voxels)
            int bad = 0;
            if (live) {
                const int64_t u = static_cast<const int64_t*>(j.a)[i], v = static_cast<const int64_t*>(j.b)[i];
                const int64_t* vbi = static_cast<const int64_t*>(j.c);
                bad = (u >= 0 && u < j.p0 && v >= 0 && v < j.p0) ? (vbi[u] != vbi[v]) : 0;
            }
            bad = (int)__popcll(__ballot(bad));
            if ((threadIdx.x & 63) == 0 && bad) atomicAdd(static_cast<int32_t*>(j.out), bad);
        } break;
        case MJ_BLKNNZ: {                              // per graph, the edges between its first and last voxel row (max)
            const int32_t* gr = static_cast<const int32_t*>(j.b);
            if (live) atomicMax(static_cast<int32_t*>(j.out), rp[gr[i + 1]] - rp[gr[i]]);
        } break;
    }
}

struct ScanJobs { int32_t* a[4]; int64_t n[4]; };
__global__ __launch_bounds__(1024) void k_scan_jobs(ScanJobs J) { d_scan_inplace(J.a[blockIdx.x], J.n[blockIdx.x], nullptr); }

// exclusive scan of deg(i), i < n, by ONE workgroup: out[0] = base, out[i+1] = base + deg(0) + .. + deg(i)
template <typename F>
__device__ void wg_scan(F deg, int n, int32_t base, int32_t* out, int32_t* total) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (n + nt - 1) / nt;
    const int b = t * per, e = min(b + per, n);
    int32_t s = 0;
    for (int i = b; i < e; ++i) s += deg(i);
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {
        int32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = base + part[t] - s;
    if (t == 0) out[0] = base;
    for (int i = b; i < e; ++i) { run += deg(i); out[i + 1] = run; }
    if (t == nt - 1 && total) *total = part[t];
    __syncthreads();
}

// stories x stories: row s lists the stories of s's graph (in the graph's CSR order), count = their number
__device__ __forceinline__ void d_build_ss(const int32_t* __restrict__ sgr_rowptr, const int32_t* __restrict__ sgr_col,
                                                   const int32_t* __restrict__ story_graph, int S, int B,
                                                   int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                   float* __restrict__ count, int32_t* __restrict__ nnz_out) {
    // a graph id out of range (reported through the story-by-graph flags) gives an empty row, never a stray read
    auto deg = [&](int s) { const int g = story_graph[s]; return (g < 0 || g >= B) ? 0 : sgr_rowptr[g + 1] - sgr_rowptr[g]; };
    wg_scan(deg, S, 0, rowptr, nnz_out);
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const int g = story_graph[s];
        if (g < 0 || g >= B) { count[s] = 0.f; continue; }
        const int b = sgr_rowptr[g], d = sgr_rowptr[g + 1] - b, o = rowptr[s];
        count[s] = (float)d;
        for (int w = 0; w < d; ++w) col[o + w] = sgr_col[b + w];
    }
}

// both aggregations of a ProgramBlock as one operator with 2 Np rows: rows [0, Np) = in-edge mean (multiplier
// 1 / max(deg, 1)); rows [Np, 2 Np) = the nodes of row i's class slot, multiplier target ratio / slot size
__device__ __forceinline__ void d_build_pp(const int32_t* __restrict__ pf_rowptr, const int32_t* __restrict__ pf_col,
                                                   const float* __restrict__ pf_count, const int32_t* __restrict__ slot_rowptr,
                                                   const int32_t* __restrict__ slot_col, const float* __restrict__ slot_count,
                                                   const int32_t* __restrict__ pcc, const float* __restrict__ ratio, int Np, int Ep,
                                                   int n_slot, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                   float* __restrict__ scale, int32_t* __restrict__ nnz_out) {
    auto deg = [&](int i) { const int k = pcc[i]; return (k < 0 || k >= n_slot) ? 0 : slot_rowptr[k + 1] - slot_rowptr[k]; };
    wg_scan(deg, Np, Ep, rowptr + Np, nnz_out);          // rowptr[Np] = Ep = pf_rowptr[Np]
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        rowptr[i] = pf_rowptr[i];
        scale[i] = __fdiv_rn(1.0f, pf_count[i]);
        const int k = pcc[i];
        if (k < 0 || k >= n_slot) { scale[Np + i] = 0.f; continue; }
        const int b = slot_rowptr[k], d = slot_rowptr[k + 1] - b, o = rowptr[Np + i];
        scale[Np + i] = __fdiv_rn(ratio[k], slot_count[k]);
        for (int w = 0; w < d; ++w) col[o + w] = slot_col[b + w];
    }
    for (int e = threadIdx.x; e < Ep; e += blockDim.x) col[e] = pf_col[e];
}

struct DerivedArgs {
    const int32_t *sgr_rowptr, *sgr_col, *story_graph; int S, B; int32_t *ss_rowptr, *ss_col; float* ss_count; int32_t* nnz_ss;
    const int32_t *pf_rowptr, *pf_col; const float* pf_count; const int32_t *slot_rowptr, *slot_col; const float* slot_count;
    const int32_t* pcc; const float* ratio; int Np, Ep, n_slot; int32_t *pp_rowptr, *pp_col; float* pp_scale; int32_t* nnz_slot;
};
// workgroup 0: stories x stories; workgroup 1: the ProgramBlock pair operator
__global__ __launch_bounds__(1024) void k_build_derived(DerivedArgs a) {
    if (blockIdx.x == 0)
        d_build_ss(a.sgr_rowptr, a.sgr_col, a.story_graph, a.S, a.B, a.ss_rowptr, a.ss_col, a.ss_count, a.nnz_ss);
    else
        d_build_pp(a.pf_rowptr, a.pf_col, a.pf_count, a.slot_rowptr, a.slot_col, a.slot_count, a.pcc, a.ratio, a.Np, a.Ep,
                   a.n_slot, a.pp_rowptr, a.pp_col, a.pp_scale, a.nnz_slot);
}

// ---------------------------------------------------------------- check_connectivity (evaluation)
// code[src * Np + dst]: bit 0 = a voxel edge with both ends unmasked joins voxels assigned to (src, dst), bit 1 =
// (src, dst) is an edge of the input program graph.  OR is order-free: the result does not depend on scheduling.
__global__ void k_conn_mark(const int64_t* __restrict__ ve, int64_t Ev, const float* __restrict__ mask,
                            const int64_t* __restrict__ prog_of_voxel, int64_t Nv, const int64_t* __restrict__ pe, int64_t Ep,
                            int64_t Np, int32_t* __restrict__ code) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < Ev) {
        const int64_t a = ve[t], b = ve[Ev + t];
        if ((uint64_t)a < (uint64_t)Nv && (uint64_t)b < (uint64_t)Nv && mask[a] + mask[b] == 2.f) {
            const int64_t pa = prog_of_voxel[a], pb = prog_of_voxel[b];
            if ((uint64_t)pa < (uint64_t)Np && (uint64_t)pb < (uint64_t)Np) atomicOr(&code[pa * Np + pb], 1);
        }
    } else if (t < Ev + Ep) {
        const int64_t k = t - Ev;
        const int64_t a = pe[k], b = pe[Ep + k];
        if ((uint64_t)a < (uint64_t)Np && (uint64_t)b < (uint64_t)Np) atomicOr(&code[a * Np + b], 2);
    }
}

// The three edge lists in ascending (src, dst) order -- what torch.unique(dim=1) returns -- by ONE workgroup: thread t
// owns a contiguous range of the code table, counts its entries per class, three threads scan the counts, every thread
// writes its entries behind its predecessors'.  class 3 = both (inter), 2 = program graph only (missing), 1 = generated only.
__global__ __launch_bounds__(1024) void k_conn_lists(const int32_t* __restrict__ code, int64_t n, int64_t Np,
                                                     int64_t* __restrict__ inter, int64_t cap_i, int64_t* __restrict__ missing,
                                                     int64_t cap_m, int64_t* __restrict__ gen, int64_t cap_g,
                                                     int64_t* __restrict__ counts) {
    __shared__ int32_t cnt[3][1024];
    __shared__ int32_t total[3];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024, lo = (int64_t)t * per, hi = lo + per < n ? lo + per : n;
    int32_t c[3] = {0, 0, 0};
    for (int64_t k = lo; k < hi; ++k) {
        const int32_t v = code[k];
        if (v) ++c[v - 1];
    }
    cnt[0][t] = c[0]; cnt[1][t] = c[1]; cnt[2][t] = c[2];
    __syncthreads();
    if (t < 3) {                                            // exclusive scan of one class's counts
        int32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const int32_t v = cnt[t][i]; cnt[t][i] = run; run += v; }
        total[t] = run;
    }
    __syncthreads();
    int64_t at[3] = {cnt[0][t], cnt[1][t], cnt[2][t]};
    int64_t* const dst[3] = {gen, missing, inter};
    const int64_t cap[3] = {cap_g, cap_m, cap_i};
    for (int64_t k = lo; k < hi; ++k) {
        const int32_t v = code[k];
        if (v) {
            const int cls = v - 1;
            const int64_t pos = at[cls]++;
            if (pos < cap[cls]) { dst[cls][pos] = k / Np; dst[cls][cap[cls] + pos] = k % Np; }
        }
    }
    if (t == 0) { counts[0] = total[2]; counts[1] = total[1]; counts[2] = total[0]; }       // inter, missing, generated
}

}  // namespace

extern "C" {

int bg_version(void) { return 1; }

const char* bg_error_string(int code) {
    switch (code) {
        case BG_OK: return "ok";
        case BG_ERR_ARG: return "bad argument (null pointer or inconsistent size)";
        case BG_ERR_SHAPE: return "shape not supported by this kernel";
        case BG_ERR_ALIGN: return "pointer or pitch not 16-byte aligned";
        case BG_ERR_LAUNCH: return "HIP launch / enqueue error";
        case BG_ERR_SCRATCH: return "request exceeds the library's static scratch buffer (use the entry point that takes a workspace)";
        default: return "unknown error";
    }
}

int bg_plan_csr(const int64_t* key, const int64_t* other, int64_t E, int64_t N,
                int32_t* rowptr, int32_t* col, int32_t* perm, int32_t* cursor, int32_t* flags,
                bgStream stream) {
    if (E < 0 || N < 0 || !rowptr || !cursor || !flags) return BG_ERR_ARG;
    if (E > 0 && (!key || !col || !perm)) return BG_ERR_ARG;        /* other == NULL: slot k points at row perm[k] */
    if (E >= (int64_t)1 << 31 || N >= ((int64_t)1 << 31) - 1) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    BG_HIP(bg_zero_async(rowptr, sizeof(int32_t) * (N + 1), s));
    if (E > 0) {
        k_hist<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(key, E, N, rowptr, flags);
        BG_LAUNCH_CHECK();
    }
    k_scan_inplace<<<1, 1024, 0, s>>>(rowptr, N + 1);
    BG_LAUNCH_CHECK();
    if (E > 0) {
        BG_HIP(bg_copy_async(cursor, rowptr, sizeof(int32_t) * (N + 1), s));
        k_place<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(key, E, N, cursor, perm, flags);
        BG_LAUNCH_CHECK();
        // cursor is free again after placement: it becomes the work list of long rows
        k_rowsort_short<<<bg_grid_for(N, 256), 256, 0, s>>>(rowptr, N, perm, flags, cursor);
        BG_LAUNCH_CHECK();
        k_rowsort_long<<<512, 256, 0, s>>>(rowptr, perm, flags, cursor);
        BG_LAUNCH_CHECK();
        k_fill_col<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(other, perm, E, col, flags, rowptr, N);
        BG_LAUNCH_CHECK();
    }
    return BG_OK;
}

int bg_plan_segptr(const int64_t* seg_id, int64_t n, int64_t nseg, int32_t* ptr, int32_t* flags,
                   bgStream stream) {
    if (n < 0 || nseg < 0 || !ptr || !flags || (n > 0 && !seg_id)) return BG_ERR_ARG;
    if (n >= (int64_t)1 << 31) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        BG_HIP(bg_zero_async(ptr, sizeof(int32_t) * (nseg + 1), s));
        return BG_OK;
    }
    k_segptr<<<bg_grid_for(n, 256), 256, 0, s>>>(seg_id, n, nseg, ptr, flags);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_plan_floor_level(const int64_t* vfc, const int64_t* vbi, int64_t n, int64_t B,
                        int32_t* first, int32_t* level, bgStream stream) {
    if (n < 0 || B <= 0 || !first || (n > 0 && (!vfc || !vbi || !level))) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    k_fill_i32<<<bg_grid_for(B, 256), 256, 0, s>>>(first, B, 0x7fffffff);
    BG_LAUNCH_CHECK();
    if (n == 0) return BG_OK;
    k_floor_min<<<bg_grid_for(n, 256), 256, 0, s>>>(vfc, vbi, n, B, first);
    BG_LAUNCH_CHECK();
    k_floor_level<<<bg_grid_for(n, 256), 256, 0, s>>>(vfc, vbi, n, B, first, level);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_plan_counts(const int32_t* rowptr, int64_t n_rows, float* count, bgStream stream) {
    if (n_rows < 0 || !rowptr || (n_rows > 0 && !count)) return BG_ERR_ARG;
    if (n_rows == 0) return BG_OK;
    k_counts<<<bg_grid_for(n_rows, 256), 256, 0, (hipStream_t)stream>>>(rowptr, n_rows, count);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

int bg_plan_chunks(const int32_t* rowptr, int64_t n_rows, int32_t rows_per_chunk,
                   int32_t* chunk_ptr, bgStream stream) {
    if (n_rows < 0 || rows_per_chunk <= 0 || !rowptr || !chunk_ptr) return BG_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    k_chunk_counts<<<bg_grid_for(n_rows + 1, 256), 256, 0, s>>>(rowptr, n_rows, rows_per_chunk, chunk_ptr);
    BG_LAUNCH_CHECK();
    k_scan_inplace<<<1, 1024, 0, s>>>(chunk_ptr, n_rows + 1);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// ---------------------------------------------------------------- the whole plan in one call
namespace {
struct PlanLayout { int64_t oi[BGP_N_I32]; int64_t of[BGP_N_F32]; int64_t ni, nf, zero_end; int64_t cursor_off[kMaxBuilds]; };

inline int64_t pad64(int64_t n) { return (n + 63) / 64 * 64; }

// rows of the 11 builds, in flag order: pf, pr, slot, sg, vfw, vrv, st, gr, sgr, cv, cp
inline void build_rows(const bgPlanIn& p, int64_t* N) {
    const int64_t r[kMaxBuilds] = {p.Np, p.Np, p.n_slot, p.B, p.Nv, p.Nv, p.S, p.B, p.B, p.Nv, p.Np};
    for (int i = 0; i < kMaxBuilds; ++i) N[i] = r[i];
}

int plan_layout(const bgPlanIn& p, PlanLayout& L) {
    if (p.Np <= 0 || p.Nv <= 0 || p.Ep < 0 || p.Ev <= 0 || p.Ec <= 0 || p.B <= 0 || p.S <= 0 || p.n_slot <= 0) return BG_ERR_SHAPE;
    if (p.S * p.S > (1 << 22) || p.Np * p.Np > (1 << 22)) return BG_ERR_SHAPE;
    // k_plan_misc, k_rowsort_short_b, k_story_graph and k_floor_* give every element a thread of its own and
    // bg_grid_for caps a grid at 2^20 workgroups of 256 threads: 2^28 elements per job is the most they cover
    constexpr int64_t kMaxJob = ((int64_t)1 << 28) - 1;
    if (p.Ev > kMaxJob || p.Ec > kMaxJob || p.Nv > kMaxJob || p.Ep > kMaxJob || p.Np > kMaxJob) return BG_ERR_SHAPE;
    int64_t len[BGP_N_I32];
    const int64_t Np = p.Np, Nv = p.Nv, Ep = p.Ep, Ev = p.Ev, Ec = p.Ec, B = p.B, S = p.S, ns = p.n_slot;
    int64_t mx = Np > ns ? Np : ns; mx = mx > Nv ? mx : Nv; mx = mx > S ? mx : S; mx = mx > Ec ? mx : Ec; mx = mx > B ? mx : B;
    len[BGP_PF_ROWPTR] = Np + 1; len[BGP_PF_COL] = Ep; len[BGP_PF_PERM] = Ep;
    len[BGP_PR_ROWPTR] = Np + 1; len[BGP_PR_COL] = Ep; len[BGP_PR_PERM] = Ep;
    len[BGP_SLOT_ROWPTR] = ns + 1; len[BGP_SLOT_COL] = Np; len[BGP_SLOT_PERM] = Np;
    len[BGP_SG_ROWPTR] = B + 1; len[BGP_SG_COL] = ns; len[BGP_SG_PERM] = ns;
    len[BGP_VFW_ROWPTR] = Nv + 1; len[BGP_VFW_COL] = Ev; len[BGP_VFW_PERM] = Ev;
    len[BGP_VRV_ROWPTR] = Nv + 1; len[BGP_VRV_COL] = Ev; len[BGP_VRV_PERM] = Ev;
    len[BGP_ST_ROWPTR] = S + 1; len[BGP_ST_COL] = Nv; len[BGP_ST_PERM] = Nv; len[BGP_ST_CHUNK] = S + 1;
    len[BGP_GR_ROWPTR] = B + 1; len[BGP_GR_COL] = Nv; len[BGP_GR_PERM] = Nv; len[BGP_GR_CHUNK] = B + 1;
    len[BGP_SGR_ROWPTR] = B + 1; len[BGP_SGR_COL] = S; len[BGP_SGR_PERM] = S; len[BGP_SGR_CHUNK] = B + 1;
    len[BGP_CV_ROWPTR] = Nv + 1; len[BGP_CV_COL] = Ec; len[BGP_CV_PERM] = Ec;
    len[BGP_CP_ROWPTR] = Np + 1; len[BGP_CP_COL] = Ec; len[BGP_CP_PERM] = Ec; len[BGP_CP_CHUNK] = Np + 1;
    len[BGP_ARANGE] = mx + 1;
    len[BGP_PCC32] = Np; len[BGP_SLOTGRAPH32] = ns; len[BGP_VFC32] = Nv; len[BGP_VBI32] = Nv; len[BGP_STORYGRAPH32] = S;
    len[BGP_CEV32] = Ec; len[BGP_CEP32] = Ec;
    len[BGP_FIRST] = B; len[BGP_LEVEL] = Nv;
    len[BGP_SS_ROWPTR] = S + 1; len[BGP_SS_COL] = S * S;
    len[BGP_PP_ROWPTR] = 2 * Np + 1; len[BGP_PP_COL] = Ep + Np * Np;
    len[BGP_FLAGS] = kMaxBuilds * 4; len[BGP_STATS] = 32;
    int64_t N[kMaxBuilds];
    build_rows(p, N);
    int64_t cur = 0;
    for (int i = 0; i < kMaxBuilds; ++i) { L.cursor_off[i] = cur; cur += pad64(N[i] + 1); }
    len[BGP_CURSOR] = cur;
    len[BGP_STORYGRAPH64] = 2 * S;
    // the pieces that must start at zero come first and are cleared by one memset; FLAGS and STATS stay adjacent
    // (the caller reads both back with one copy)
    static const int zeroed[] = {BGP_PF_ROWPTR, BGP_PR_ROWPTR, BGP_SLOT_ROWPTR, BGP_SG_ROWPTR, BGP_VFW_ROWPTR, BGP_VRV_ROWPTR,
                                 BGP_ST_ROWPTR, BGP_GR_ROWPTR, BGP_SGR_ROWPTR, BGP_CV_ROWPTR, BGP_CP_ROWPTR,
                                 BGP_STORYGRAPH32, BGP_STORYGRAPH64, BGP_FLAGS, BGP_STATS};
    bool placed[BGP_N_I32] = {};
    int64_t o = 0;
    for (int k : zeroed) { L.oi[k] = o; o += pad64(len[k]); placed[k] = true; }
    L.zero_end = L.oi[BGP_STATS];
    for (int i = 0; i < BGP_N_I32; ++i)
        if (!placed[i]) { L.oi[i] = o; o += pad64(len[i]); }
    L.ni = o;
    const int64_t lf[BGP_N_F32] = {Np, ns, Nv, S, B, S, 2 * Np, Nv, Np};
    o = 0;
    for (int i = 0; i < BGP_N_F32; ++i) { L.of[i] = o; o += pad64(lf[i]); }
    L.nf = o;
    return BG_OK;
}
}  // namespace

int bg_plan_layout(const bgPlanIn* in, int64_t* offs_i32, int64_t* offs_f32, int64_t* total) {
    if (!in || !offs_i32 || !offs_f32 || !total) return BG_ERR_ARG;
    PlanLayout L;
    int rc = plan_layout(*in, L);
    if (rc != BG_OK) return rc;
    for (int i = 0; i < BGP_N_I32; ++i) offs_i32[i] = L.oi[i];
    for (int i = 0; i < BGP_N_F32; ++i) offs_f32[i] = L.of[i];
    total[0] = L.ni; total[1] = L.nf;
    return BG_OK;
}

int bg_plan_build(const bgPlanIn* in, int32_t* ai, float* af, bgStream stream) {
    if (!in || !ai || !af) return BG_ERR_ARG;
    const bgPlanIn& p = *in;
    if (!p.pe_src || !p.pe_dst || !p.ve_src || !p.ve_dst || !p.cep || !p.cev || !p.vfc || !p.vbi || !p.pcc || !p.slot_graph ||
        !p.target_ratio) return BG_ERR_ARG;
    for (int i = 0; i < 3; ++i) if (p.chunk_rows[i] <= 0) return BG_ERR_ARG;
    if (p.chunk_rows[3] < 0) return BG_ERR_ARG;
    if (!bg_aligned16(ai)) return BG_ERR_ALIGN;
    PlanLayout L;
    int rc = plan_layout(p, L);
    if (rc != BG_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    auto I = [&](int k) { return ai + L.oi[k]; };
    auto F = [&](int k) { return af + L.of[k]; };
    int32_t* flags = I(BGP_FLAGS);
    int32_t* stats = I(BGP_STATS);
    int64_t* sg64 = reinterpret_cast<int64_t*>(I(BGP_STORYGRAPH64));
    const int64_t Np = p.Np, Nv = p.Nv, Ep = p.Ep, Ev = p.Ev, Ec = p.Ec, B = p.B, S = p.S, ns = p.n_slot;
    auto g1 = [](int64_t n) { return bg_grid_for(n, 256); };

    // 1. everything that starts at zero (row pointers, story -> graph tables, flags)
    BG_HIP(bg_zero_async(ai, sizeof(int32_t) * L.zero_end, s));
    // 2. story -> graph (the key of build 8) and the first-floor table
    k_story_graph<<<g1(Nv > B ? Nv : B), 256, 0, s>>>(p.vfc, p.vbi, Nv, S, sg64, I(BGP_STORYGRAPH32), I(BGP_FIRST), B);
    BG_LAUNCH_CHECK();
    // 3. the eleven CSR builds in lockstep: six launches, blockIdx.y = the build
    CsrJobs J;
    const int64_t* keys[kMaxBuilds] = {p.pe_dst, p.pe_src, p.pcc, p.slot_graph, p.ve_dst, p.ve_src, p.vfc, p.vbi, sg64, p.cev, p.cep};
    const int64_t* others[kMaxBuilds] = {p.pe_src, p.pe_dst, nullptr, nullptr, p.ve_src, p.ve_dst, nullptr, nullptr, nullptr, p.cep, p.cev};
    const int64_t Es[kMaxBuilds] = {Ep, Ep, Np, ns, Ev, Ev, Nv, Nv, S, Ec, Ec};
    const int rps[kMaxBuilds] = {BGP_PF_ROWPTR, BGP_PR_ROWPTR, BGP_SLOT_ROWPTR, BGP_SG_ROWPTR, BGP_VFW_ROWPTR, BGP_VRV_ROWPTR,
                                 BGP_ST_ROWPTR, BGP_GR_ROWPTR, BGP_SGR_ROWPTR, BGP_CV_ROWPTR, BGP_CP_ROWPTR};
    const int cols[kMaxBuilds] = {BGP_PF_COL, BGP_PR_COL, BGP_SLOT_COL, BGP_SG_COL, BGP_VFW_COL, BGP_VRV_COL, BGP_ST_COL, BGP_GR_COL,
                                  BGP_SGR_COL, BGP_CV_COL, BGP_CP_COL};
    int64_t Ns[kMaxBuilds];
    build_rows(p, Ns);
    int64_t e_max = 1, n_max = 1;
    for (int i = 0; i < kMaxBuilds; ++i) {
        J.j[i] = CsrJob{keys[i], others[i], Es[i], Ns[i], I(rps[i]), I(cols[i]), I(cols[i] + 1), I(BGP_CURSOR) + L.cursor_off[i],
                        flags + 4 * i};
        e_max = Es[i] > e_max ? Es[i] : e_max;
        n_max = Ns[i] > n_max ? Ns[i] : n_max;
    }
    const dim3 ge(bg_grid_for(e_max, 256, 2048), kMaxBuilds), gn(g1(n_max), kMaxBuilds);
    k_hist_b<<<ge, 256, 0, s>>>(J, stats);
    BG_LAUNCH_CHECK();
    k_scan_b<<<kMaxBuilds, 1024, 0, s>>>(J);
    BG_LAUNCH_CHECK();
    k_place_b<<<ge, 256, 0, s>>>(J);
    BG_LAUNCH_CHECK();
    k_rowsort_short_b<<<gn, 256, 0, s>>>(J);
    BG_LAUNCH_CHECK();
    k_rowsort_long_b<<<dim3(128, kMaxBuilds), 256, 0, s>>>(J);
    BG_LAUNCH_CHECK();
    k_fill_col_b<<<ge, 256, 0, s>>>(J);
    BG_LAUNCH_CHECK();
    // 4. floor level of every voxel inside its own building
    k_floor_min<<<g1(Nv), 256, 0, s>>>(p.vfc, p.vbi, Nv, B, I(BGP_FIRST));
    BG_LAUNCH_CHECK();
    k_floor_level<<<g1(Nv), 256, 0, s>>>(p.vfc, p.vbi, Nv, B, I(BGP_FIRST), I(BGP_LEVEL));
    BG_LAUNCH_CHECK();
    // 5. int32 copies, counts, chunk counts, statistics: one launch
    MiscJobs M;
    int nm = 0;
    int64_t m_max = 1;
    auto job = [&](int kind, const void* a, const void* b, const void* c, void* out, int64_t n, int p0 = 0, int p1 = 0) {
        M.j[nm++] = MiscJob{a, b, c, out, n, kind, p0, p1, 0};
        m_max = n + 1 > m_max ? n + 1 : m_max;
    };
    int64_t mx = Np > ns ? Np : ns; mx = mx > Nv ? mx : Nv; mx = mx > S ? mx : S; mx = mx > Ec ? mx : Ec; mx = mx > B ? mx : B;
    job(MJ_ARANGE, nullptr, nullptr, nullptr, I(BGP_ARANGE), mx + 1);
    job(MJ_I64_I32, p.pcc, nullptr, nullptr, I(BGP_PCC32), Np);
    job(MJ_I64_I32, p.slot_graph, nullptr, nullptr, I(BGP_SLOTGRAPH32), ns);
    job(MJ_I64_I32, p.vfc, nullptr, nullptr, I(BGP_VFC32), Nv);
    job(MJ_I64_I32, p.vbi, nullptr, nullptr, I(BGP_VBI32), Nv);
    job(MJ_I64_I32, p.cev, nullptr, nullptr, I(BGP_CEV32), Ec);
    job(MJ_I64_I32, p.cep, nullptr, nullptr, I(BGP_CEP32), Ec);
    job(MJ_COUNTS, I(BGP_PF_ROWPTR), nullptr, nullptr, F(BGP_PF_COUNT), Np);
    job(MJ_COUNTS, I(BGP_SLOT_ROWPTR), nullptr, nullptr, F(BGP_SLOT_COUNT), ns);
    job(MJ_COUNTS, I(BGP_VFW_ROWPTR), nullptr, nullptr, F(BGP_VFW_COUNT), Nv);
    job(MJ_COUNTS, I(BGP_ST_ROWPTR), nullptr, nullptr, F(BGP_ST_COUNT), S);
    job(MJ_COUNTS, I(BGP_SGR_ROWPTR), nullptr, nullptr, F(BGP_SGR_COUNT), B);
    job(MJ_HAS, I(BGP_VFW_ROWPTR), nullptr, nullptr, F(BGP_VOX_HAS), Nv);
    job(MJ_HAS, I(BGP_PF_ROWPTR), nullptr, nullptr, F(BGP_PROG_HAS), Np);
    job(MJ_CHUNKCNT, I(BGP_ST_ROWPTR), nullptr, nullptr, I(BGP_ST_CHUNK), S, p.chunk_rows[0]);
    job(MJ_CHUNKCNT, I(BGP_GR_ROWPTR), nullptr, nullptr, I(BGP_GR_CHUNK), B, p.chunk_rows[1]);
    job(MJ_CHUNKCNT, I(BGP_SGR_ROWPTR), nullptr, nullptr, I(BGP_SGR_CHUNK), B, p.chunk_rows[2]);
    if (p.chunk_rows[3] > 0) job(MJ_CHUNKCNT, I(BGP_CP_ROWPTR), nullptr, nullptr, I(BGP_CP_CHUNK), Np, p.chunk_rows[3]);
    job(MJ_MAXI32, I(BGP_LEVEL), nullptr, nullptr, stats + BGS_MAX_LEVEL, Nv);
    job(MJ_ROWLEN, I(BGP_VFW_ROWPTR), nullptr, nullptr, stats, Nv, BGS_MIN_VDEG, BGS_VOX_MAX_DEG);
    job(MJ_ROWLEN, I(BGP_VRV_ROWPTR), nullptr, nullptr, stats, Nv, -1, BGS_VOX_MAX_DEG);
    job(MJ_ROWLEN, I(BGP_PF_ROWPTR), nullptr, nullptr, stats, Np, BGS_MIN_PDEG, -1);
    job(MJ_ROWLEN, I(BGP_CV_ROWPTR), nullptr, nullptr, stats, Nv, BGS_MIN_CROSS, BGS_MAXLEN_CV);
    job(MJ_ROWLEN, I(BGP_ST_ROWPTR), nullptr, nullptr, stats, S, BGS_MIN_STORY, BGS_MAXLEN_ST);
    job(MJ_ROWLEN, I(BGP_GR_ROWPTR), nullptr, nullptr, stats, B, -1, BGS_MAXLEN_GR);
    job(MJ_ROWLEN, I(BGP_SGR_ROWPTR), nullptr, nullptr, stats, B, -1, BGS_MAXLEN_SGR);
    job(MJ_ROWLEN, I(BGP_CP_ROWPTR), nullptr, nullptr, stats, Np, -1, BGS_MAXLEN_CP);
    job(MJ_LEAVING, p.ve_src, p.ve_dst, p.vbi, stats + BGS_N_LEAVING, Ev, (int)Nv);
    job(MJ_BLKNNZ, I(BGP_VFW_ROWPTR), I(BGP_GR_ROWPTR), nullptr, stats + BGS_BLK_NNZ, B);
    static_assert(kMaxMisc >= 29, "job table");
    k_plan_misc<<<dim3(g1(m_max), nm), 256, 0, s>>>(M);
    BG_LAUNCH_CHECK();
    // 6. chunk tables: scans of the chunk counts
    ScanJobs SJ;
    int nsj = 0;
    SJ.a[nsj] = I(BGP_ST_CHUNK); SJ.n[nsj++] = S + 1;
    SJ.a[nsj] = I(BGP_GR_CHUNK); SJ.n[nsj++] = B + 1;
    SJ.a[nsj] = I(BGP_SGR_CHUNK); SJ.n[nsj++] = B + 1;
    if (p.chunk_rows[3] > 0) { SJ.a[nsj] = I(BGP_CP_CHUNK); SJ.n[nsj++] = Np + 1; }
    for (int i = nsj; i < 4; ++i) { SJ.a[i] = nullptr; SJ.n[i] = 0; }
    k_scan_jobs<<<nsj, 1024, 0, s>>>(SJ);
    BG_LAUNCH_CHECK();
    // 7. the two derived operators
    DerivedArgs D{I(BGP_SGR_ROWPTR), I(BGP_SGR_COL), I(BGP_STORYGRAPH32), (int)S, (int)B, I(BGP_SS_ROWPTR), I(BGP_SS_COL),
                  F(BGP_SS_COUNT), stats + BGS_NNZ_SS,
                  I(BGP_PF_ROWPTR), I(BGP_PF_COL), F(BGP_PF_COUNT), I(BGP_SLOT_ROWPTR), I(BGP_SLOT_COL), F(BGP_SLOT_COUNT),
                  I(BGP_PCC32), p.target_ratio, (int)Np, (int)Ep, (int)ns, I(BGP_PP_ROWPTR), I(BGP_PP_COL), F(BGP_PP_SCALE),
                  stats + BGS_NNZ_SLOT};
    k_build_derived<<<2, 1024, 0, s>>>(D);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

// check_connectivity (util_eval.py:102-136) as a code table and three sorted edge lists: see include/bgan.h
int bg_connectivity(const int64_t* voxel_edge, int64_t Ev, const float* mask, const int64_t* prog_of_voxel, int64_t Nv,
                    const int64_t* program_edge, int64_t Ep, int64_t Np, int32_t* code, int64_t* inter, int64_t cap_inter,
                    int64_t* missing, int64_t cap_missing, int64_t* gen, int64_t cap_gen, int64_t* counts, bgStream stream) {
    if (Ev < 0 || Ep < 0 || Nv < 0 || Np <= 0 || cap_inter < 0 || cap_missing < 0 || cap_gen < 0) return BG_ERR_ARG;
    if (!code || !counts || (Ev > 0 && (!voxel_edge || !mask || !prog_of_voxel)) || (Ep > 0 && !program_edge) ||
        (cap_inter > 0 && !inter) || (cap_missing > 0 && !missing) || (cap_gen > 0 && !gen)) return BG_ERR_ARG;
    if (Np > (1 << 13) || Ev + Ep > ((int64_t)1 << 40)) return BG_ERR_SHAPE;            // the table has Np^2 <= 2^26 entries
    const int64_t n = Np * Np;
    hipStream_t s = (hipStream_t)stream;
    int rc = bg_zero_bytes(code, n * (int64_t)sizeof(int32_t), stream);
    if (rc != BG_OK) return rc;
    if (Ev + Ep > 0) {
        k_conn_mark<<<bg_grid_for(Ev + Ep, 256, 0x7fffffff), 256, 0, s>>>(voxel_edge, Ev, mask, prog_of_voxel, Nv, program_edge, Ep,
                                                                           Np, code);
        BG_LAUNCH_CHECK();
    }
    k_conn_lists<<<1, 1024, 0, s>>>(code, n, Np, inter, cap_inter, missing, cap_missing, gen, cap_gen, counts);
    BG_LAUNCH_CHECK();
    return BG_OK;
}

}  // extern "C"

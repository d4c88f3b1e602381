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

__global__ void k_hist(const int64_t* __restrict__ key, int64_t E, int64_t N,
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

// In-place inclusive scan of a[0..n) by ONE workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_scan_inplace(int32_t* __restrict__ a, int64_t n) {
    __shared__ int32_t part[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = (int64_t)t * per;
    const int64_t e = (b + per < n) ? b + per : n;
    int32_t s = 0;
    for (int64_t i = b; i < e; ++i) s += a[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int32_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int32_t run = part[t] - s;
    for (int64_t i = b; i < e; ++i) {
        run += a[i];
        a[i] = run;
    }
}

__global__ void k_place(const int64_t* __restrict__ key, int64_t E, int64_t N,
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
__global__ void k_rowsort_short(const int32_t* __restrict__ rowptr, int64_t N, int32_t* __restrict__ perm,
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

// workgroups walk the work list; one row at a time per workgroup (LDS when it fits)
__global__ __launch_bounds__(256) void k_rowsort_long(const int32_t* __restrict__ rowptr, int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ flags,
                                                      const int32_t* __restrict__ worklist) {
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

__global__ void k_fill_col(const int64_t* __restrict__ other, const int32_t* __restrict__ perm, int64_t E,
                           int32_t* __restrict__ col, int32_t* __restrict__ flags) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int moved = 0;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < E; k += stride) {
        int32_t p = perm[k];
        col[k] = (int32_t)other[p];
        moved += (p != (int32_t)k);
    }
    if (moved) atomicAdd(&flags[1], moved);
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
    if (E > 0 && (!key || !other || !col || !perm)) return BG_ERR_ARG;
    if (E >= (int64_t)1 << 31 || N >= ((int64_t)1 << 31) - 1) return BG_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    BG_HIP(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (N + 1), s));
    if (E > 0) {
        k_hist<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(key, E, N, rowptr, flags);
        BG_LAUNCH_CHECK();
    }
    k_scan_inplace<<<1, 1024, 0, s>>>(rowptr, N + 1);
    BG_LAUNCH_CHECK();
    if (E > 0) {
        BG_HIP(hipMemcpyAsync(cursor, rowptr, sizeof(int32_t) * (N + 1), hipMemcpyDeviceToDevice, s));
        k_place<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(key, E, N, cursor, perm, flags);
        BG_LAUNCH_CHECK();
        // cursor is free again after placement: it becomes the work list of long rows
        k_rowsort_short<<<bg_grid_for(N, 256), 256, 0, s>>>(rowptr, N, perm, flags, cursor);
        BG_LAUNCH_CHECK();
        k_rowsort_long<<<512, 256, 0, s>>>(rowptr, perm, flags, cursor);
        BG_LAUNCH_CHECK();
        k_fill_col<<<bg_grid_for(E, 256, 2048), 256, 0, s>>>(other, perm, E, col, flags);
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
        BG_HIP(hipMemsetAsync(ptr, 0, sizeof(int32_t) * (nseg + 1), s));
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

}  // extern "C"

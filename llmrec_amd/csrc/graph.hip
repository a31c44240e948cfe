// graph.hip - R1: device-side graph ingest (COO -> CSR), degree scales, SpMM long-row plan.
// Replaces the host scipy normalisation + COO tensor of reference main.py:84-93,114-134.
// These run once per graph (set-up), so the sort is rocPRIM's radix sort via hipCUB; the
// hand-written kernels are the ones on the per-step path (spmm.hip etc.).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace llmrec {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ void pack_keys_kernel(int64_t nnz, const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                 uint64_t* __restrict__ keys) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x)
        keys[j] = ((uint64_t)row[j] << 32) | (uint64_t)(uint32_t)col[j];
}

// sorted keys -> colidx and rowptr. Position j opens every row in (row(j-1), row(j)].
__global__ void unpack_keys_kernel(int64_t nnz, int64_t n_rows, const uint64_t* __restrict__ keys,
                                   int32_t* __restrict__ colidx, int32_t* __restrict__ rowptr) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= nnz; j += stride) {
        int64_t r_prev = (j == 0) ? -1 : (int64_t)(keys[j - 1] >> 32);
        int64_t r_cur = (j == nnz) ? n_rows : (int64_t)(keys[j] >> 32);
        if (j < nnz) colidx[j] = (int32_t)(uint32_t)(keys[j] & 0xffffffffull);
        for (int64_t r = r_prev + 1; r <= r_cur; ++r) rowptr[r] = (int32_t)j;
    }
}

__global__ void degree_scale_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, float* __restrict__ scale) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        int32_t deg = rowptr[r + 1] - rowptr[r];
        // reference main.py:115-117: np.power(rowsum + 1e-8, -0.5), inf -> 0, on fp32 row sums.
        // In fp32, deg + 1e-8 == deg for deg >= 1; deg == 0 gives 1e-8^-0.5 = 1e4 in the
        // reference, multiplied into an empty row (no effect), so 0 is equivalent.
        scale[r] = deg > 0 ? (float)(1.0 / sqrt((double)deg)) : 0.0f;
    }
}

__global__ void row_constant_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const float* __restrict__ val,
                                    float* __restrict__ row_const, int32_t* __restrict__ not_const) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        int32_t s = rowptr[r], e = rowptr[r + 1];
        float c = (e > s) ? val[s] : 0.0f;
        bool bad = false;
        for (int32_t j = s + 1; j < e; ++j) bad |= (val[j] != c);
        row_const[r] = c;
        if (bad) atomicOr(not_const, 1);
    }
}

__global__ void flag_finish_kernel(int32_t* flag) { flag[0] = flag[0] ? 0 : 1; }

// row classes of the SpMM (include/llmrec_hip.h): 0 lane group, 1 wavefront, 2 block, 3 split into segments
__device__ __forceinline__ int row_class(int32_t deg, int32_t t_wave, int32_t t_block) {
    return deg <= LLMREC_SPMM_LONG_ROW ? 0 : (deg <= t_wave ? 1 : (deg <= t_block ? 2 : 3));
}

__global__ void plan_count_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, int32_t t_wave, int32_t t_block, int32_t segment,
                                  int32_t* __restrict__ counts) {
    int32_t n1 = 0, n2 = 0, n3 = 0, ns = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t deg = rowptr[r + 1] - rowptr[r];
        const int c = row_class(deg, t_wave, t_block);
        n1 += c == 1; n2 += c == 2;
        if (c == 3) { n3 += 1; ns += (deg + segment - 1) / segment; }
    }
    if (n1) atomicAdd(&counts[0], n1);
    if (n2) atomicAdd(&counts[1], n2);
    if (n3) { atomicAdd(&counts[2], n3); atomicAdd(&counts[3], ns); }
}

__global__ void plan_fill_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, int32_t t_wave, int32_t t_block, int32_t segment,
                                 int32_t* __restrict__ cursors,
                                 int32_t* __restrict__ wave_rows, int32_t* __restrict__ block_rows,
                                 int32_t* __restrict__ split_rows, int32_t* __restrict__ split_seg_begin,
                                 int32_t* __restrict__ seg_split) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t deg = rowptr[r + 1] - rowptr[r];
        const int c = row_class(deg, t_wave, t_block);
        if (c == 1) wave_rows[atomicAdd(&cursors[0], 1)] = (int32_t)r;
        else if (c == 2) block_rows[atomicAdd(&cursors[1], 1)] = (int32_t)r;
        else if (c == 3) {
            const int32_t k = (deg + segment - 1) / segment;
            const int32_t slot = atomicAdd(&cursors[2], 1);
            const int32_t base = atomicAdd(&cursors[3], k);
            split_rows[slot] = (int32_t)r;
            split_seg_begin[slot] = base;
            for (int32_t s = 0; s < k; ++s) seg_split[base + s] = slot;
        }
    }
}

// keys (id << 32 | j) of the rows to scatter; ids < 0 sort to the end (key = all ones) and are skipped
__global__ void scatter_keys_kernel(int64_t n, const int64_t* __restrict__ ids, uint64_t* __restrict__ keys) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        keys[j] = ids[j] < 0 ? ~0ull : (((uint64_t)ids[j] << 32) | (uint64_t)(uint32_t)j);
}

// The per-step case (n <= 16 384: the gradient rows of every rank's batch, 2 B x world) sorts its keys in ONE block: bitonic network over
// the keys in LDS (128 KB at most) - no vendor library on the per-step path of the row-sharded step; larger n goes
// through rocPRIM's radix sort below. Keys are distinct ((id, j) pairs; ~0 for skipped / padded slots), so the result is THE sorted order.
constexpr int SCATTER_SORT_MAX = 16384;
__global__ __launch_bounds__(1024) void scatter_sort_block_kernel(int n, int n_pow2, const int64_t* __restrict__ ids, uint64_t* __restrict__ keys_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* k = reinterpret_cast<uint64_t*>(smem_raw);
    for (int i = threadIdx.x; i < n_pow2; i += 1024)
        k[i] = (i < n && ids[i] >= 0) ? (((uint64_t)ids[i] << 32) | (uint64_t)(uint32_t)i) : ~0ull;
    __syncthreads();
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (n_pow2 >> 1); t += 1024) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = k[i], b = k[j];
                if ((a > b) == up) { k[i] = b; k[j] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n; i += 1024) keys_out[i] = k[i];
}

// distinct ids >= 0, ascending, in one block (llmrec_sort_unique_ids_i32): bitonic network over 32-bit keys in LDS (skipped / padded slots
// = 0xffffffff sort to the end), heads of runs counted per thread chunk, block-wide exclusive scan, ordered write
__global__ __launch_bounds__(1024) void sort_unique_block_kernel(int n, int n_pow2, const int64_t* __restrict__ ids, int32_t* __restrict__ list,
                                                                int32_t* __restrict__ n_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* k = reinterpret_cast<uint32_t*>(smem_raw);
    __shared__ int32_t wave_total[16];
    for (int i = threadIdx.x; i < n_pow2; i += 1024)
        k[i] = (i < n && ids[i] >= 0 && ids[i] < 0xffffffffll) ? (uint32_t)ids[i] : 0xffffffffu;
    __syncthreads();
    for (int size = 2; size <= n_pow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (n_pow2 >> 1); t += 1024) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint32_t a = k[i], b = k[j];
                if ((a > b) == up) { k[i] = b; k[j] = a; }
            }
            __syncthreads();
        }
    }
    const int per = (n_pow2 + 1023) / 1024;                              // contiguous chunk per thread
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, n_pow2);
    int32_t c = 0;
    for (int i = i0; i < i1; ++i) c += k[i] != 0xffffffffu && (i == 0 || k[i - 1] != k[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int32_t base = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) { const int32_t t = wave_total[w2]; if (w2 < wave) base += t; total += t; }
    int32_t o = base + incl - c;
    for (int i = i0; i < i1; ++i)
        if (k[i] != 0xffffffffu && (i == 0 || k[i - 1] != k[i])) list[o++] = (int32_t)k[i];
    for (int i = total + threadIdx.x; i < n; i += 1024) list[i] = 0;     // the unused tail: defined values
    if (threadIdx.x == 0) *n_out = total;
}

// one 16-lane group per sorted position; the head of a run of equal ids adds the run's rows in ascending j
__global__ __launch_bounds__(256) void scatter_runs_kernel(int64_t n, const uint64_t* __restrict__ keys, const float* __restrict__ rows,
                                                           int64_t ldr, int d, float alpha, float* __restrict__ dst, int64_t ldd) {
    const int gl = threadIdx.x & 15;
    const int64_t j0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j0 >= n) return;
    const uint64_t k0 = keys[j0];
    if (k0 == ~0ull) return;
    const uint64_t id = k0 >> 32;
    if (j0 > 0 && (keys[j0 - 1] >> 32) == id) return;             // not the head of its run
    float* out = dst + (int64_t)id * ldd;
    for (int c = gl; c < d; c += 16) {
        float s = out[c];
        for (int64_t j = j0; j < n; ++j) {
            const uint64_t k = keys[j];
            if ((k >> 32) != id) break;
            s = fmaf(alpha, rows[(int64_t)(uint32_t)k * ldr + c], s);
        }
        out[c] = s;
    }
}

// one 16-lane group per destination row: copies the row's index segment
__global__ __launch_bounds__(256) void csr_permute_rows_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                               const int32_t* __restrict__ perm, const int32_t* __restrict__ p_rowptr,
                                                               int32_t* __restrict__ p_colidx) {
    const int gl = threadIdx.x & 15;
    for (int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); s < n_rows; s += (int64_t)gridDim.x * 16) {
        const int32_t src = rowptr[perm[s]], dst = p_rowptr[s], len = p_rowptr[s + 1] - dst;
        for (int32_t j = gl; j < len; j += 16) p_colidx[dst + j] = colidx[src + j];
    }
}

}  // namespace llmrec

using namespace llmrec;

extern "C" {

int llmrec_abi_version(void) { return LLMREC_ABI_VERSION; }

const char* llmrec_status_string(int status) {
    switch (status) {
        case LLMREC_OK: return "ok";
        case LLMREC_EINVAL: return "invalid argument";
        case LLMREC_EHIP: return "HIP runtime error";
        case LLMREC_EWORKSPACE: return "workspace too small";
        case LLMREC_EUNSUPPORTED: return "unsupported shape";
        default: return "unknown status";
    }
}

const char* llmrec_last_error(void) { return g_err; }

int64_t llmrec_csr_build_workspace_bytes(int64_t n_rows, int64_t nnz) {
    (void)n_rows;
    if (nnz < 0) return -1;
    // two key buffers (8 B each) + alternate value buffer (4 B) + rocPRIM temp storage bound
    return align_up(8 * nnz, 256) * 2 + align_up(4 * nnz, 256) + align_up(nnz + (32ll << 20), 256);
}

int llmrec_csr_build(int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int64_t* coo_row, const int64_t* coo_col, const float* coo_val,
                     int32_t* rowptr, int32_t* colidx, float* val,
                     void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "csr_build: negative size");
    LLMREC_CHECK_ARG(n_rows < (1ll << 31) && n_cols < (1ll << 32) && nnz < (1ll << 31),
                     "csr_build: sizes exceed the int32 CSR range (rows %lld cols %lld nnz %lld)",
                     (long long)n_rows, (long long)n_cols, (long long)nnz);
    LLMREC_CHECK_ARG(rowptr && (nnz == 0 || (coo_row && coo_col && colidx)), "csr_build: null pointer");
    LLMREC_CHECK_ARG((coo_val == nullptr) == (val == nullptr), "csr_build: coo_val and val must both be set or both NULL");
    if (workspace_bytes < llmrec_csr_build_workspace_bytes(n_rows, nnz) || (nnz > 0 && !workspace)) {
        set_error("csr_build: workspace %lld < %lld", (long long)workspace_bytes,
                  (long long)llmrec_csr_build_workspace_bytes(n_rows, nnz));
        return LLMREC_EWORKSPACE;
    }
    char* ws = (char*)workspace;
    uint64_t* keys_a = (uint64_t*)ws;
    uint64_t* keys_b = (uint64_t*)(ws + align_up(8 * nnz, 256));
    float* val_alt = (float*)(ws + 2 * align_up(8 * nnz, 256));
    void* temp = ws + 2 * align_up(8 * nnz, 256) + align_up(4 * nnz, 256);
    size_t temp_avail = (size_t)align_up(nnz + (32ll << 20), 256);

    if (nnz > 0) {
        pack_keys_kernel<<<grid_for(nnz, 256), 256, 0, stream>>>(nnz, coo_row, coo_col, keys_a);
        LLMREC_LAUNCH_CHECK();
        int row_bits = 1;
        while ((1ll << row_bits) < n_rows) ++row_bits;
        int end_bit = 32 + row_bits;
        hipcub::DoubleBuffer<uint64_t> kbuf(keys_a, keys_b);
        size_t need = 0;
        if (coo_val) {
            LLMREC_HIP(hipMemcpyAsync(val, coo_val, 4 * nnz, hipMemcpyDeviceToDevice, stream));
            hipcub::DoubleBuffer<float> vbuf(val, val_alt);
            LLMREC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, kbuf, vbuf, (int)nnz, 0, end_bit, stream));
            if (need > temp_avail) { set_error("csr_build: rocPRIM needs %zu B temp > %zu", need, temp_avail); return LLMREC_EWORKSPACE; }
            LLMREC_HIP(hipcub::DeviceRadixSort::SortPairs(temp, need, kbuf, vbuf, (int)nnz, 0, end_bit, stream));
            if (vbuf.Current() != val)
                LLMREC_HIP(hipMemcpyAsync(val, vbuf.Current(), 4 * nnz, hipMemcpyDeviceToDevice, stream));
        } else {
            LLMREC_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need, kbuf, (int)nnz, 0, end_bit, stream));
            if (need > temp_avail) { set_error("csr_build: rocPRIM needs %zu B temp > %zu", need, temp_avail); return LLMREC_EWORKSPACE; }
            LLMREC_HIP(hipcub::DeviceRadixSort::SortKeys(temp, need, kbuf, (int)nnz, 0, end_bit, stream));
        }
        unpack_keys_kernel<<<grid_for(nnz + 1, 256), 256, 0, stream>>>(nnz, n_rows, kbuf.Current(), colidx, rowptr);
        LLMREC_LAUNCH_CHECK();
    } else {
        LLMREC_HIP(hipMemsetAsync(rowptr, 0, 4 * (n_rows + 1), stream));
    }
    return LLMREC_OK;
}

int llmrec_degree_scale(int64_t n_rows, const int32_t* rowptr, float* scale, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_rows >= 0 && rowptr && (scale || n_rows == 0), "degree_scale: bad argument");
    if (n_rows == 0) return LLMREC_OK;
    degree_scale_kernel<<<grid_for(n_rows, 256), 256, 0, (hipStream_t)stream_>>>(n_rows, rowptr, scale);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_csr_row_constant(int64_t n_rows, const int32_t* rowptr, const float* val,
                            float* row_const, int32_t* flag_out, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && rowptr && flag_out && (n_rows == 0 || (val && row_const)), "csr_row_constant: bad argument");
    LLMREC_HIP(hipMemsetAsync(flag_out, 0, 4, stream));
    if (n_rows > 0) {
        row_constant_kernel<<<grid_for(n_rows, 256), 256, 0, stream>>>(n_rows, rowptr, val, row_const, flag_out);
        LLMREC_LAUNCH_CHECK();
    }
    flag_finish_kernel<<<1, 1, 0, stream>>>(flag_out);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_csr_permute_rows(int64_t n_rows, const int32_t* rowptr, const int32_t* colidx, const int32_t* perm, const int32_t* p_rowptr,
                            int32_t* p_colidx, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n_rows >= 0 && rowptr && perm && p_rowptr, "csr_permute_rows: bad argument");
    if (n_rows == 0) return LLMREC_OK;
    csr_permute_rows_kernel<<<grid_for(n_rows, 16, 256 * 32), 256, 0, (hipStream_t)stream_>>>(n_rows, rowptr, colidx, perm, p_rowptr, p_colidx);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int llmrec_spmm_plan_count(int64_t n_rows, const int32_t* rowptr, int32_t t_wave, int32_t t_block, int32_t segment,
                           int32_t* scratch4, int32_t* counts_host, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && rowptr && scratch4 && counts_host, "spmm_plan_count: bad argument");
    LLMREC_CHECK_ARG(t_wave >= LLMREC_SPMM_LONG_ROW && t_block >= t_wave && segment >= LLMREC_SPMM_LONG_ROW, "spmm_plan_count: bad thresholds");
    LLMREC_HIP(hipMemsetAsync(scratch4, 0, 16, stream));
    if (n_rows > 0) {
        plan_count_kernel<<<grid_for(n_rows, 256), 256, 0, stream>>>(n_rows, rowptr, t_wave, t_block, segment, scratch4);
        LLMREC_LAUNCH_CHECK();
    }
    LLMREC_HIP(hipMemcpyAsync(counts_host, scratch4, 16, hipMemcpyDeviceToHost, stream));
    LLMREC_HIP(hipStreamSynchronize(stream));
    return LLMREC_OK;
}

int llmrec_spmm_plan_fill(int64_t n_rows, const int32_t* rowptr, int32_t t_wave, int32_t t_block, int32_t segment, int32_t* scratch4,
                          int32_t* wave_rows, int32_t* block_rows, int32_t* split_rows,
                          int32_t* split_seg_begin, int32_t* seg_split, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n_rows >= 0 && rowptr && scratch4, "spmm_plan_fill: bad argument");
    LLMREC_CHECK_ARG(t_wave >= LLMREC_SPMM_LONG_ROW && t_block >= t_wave && segment >= LLMREC_SPMM_LONG_ROW, "spmm_plan_fill: bad thresholds");
    LLMREC_HIP(hipMemsetAsync(scratch4, 0, 16, stream));
    if (n_rows > 0) {
        plan_fill_kernel<<<grid_for(n_rows, 256), 256, 0, stream>>>(n_rows, rowptr, t_wave, t_block, segment, scratch4, wave_rows, block_rows, split_rows,
                                                                    split_seg_begin, seg_split);
        LLMREC_LAUNCH_CHECK();
    }
    return LLMREC_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize of kernel `which` on the CURRENT device, set once per (kernel, device); false = refused
static bool big_lds_ok(int which, const void* fn, int bytes) {
    static signed char state[2][64] = {};                      // 0 unknown, 1 granted, -1 refused (a race between two first calls sets the same value twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
    if (state[which][dev] == 0) {
        const bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[which][dev] = ok ? 1 : -1;
    }
    return state[which][dev] > 0;
}

int llmrec_sort_unique_ids_i32(int64_t n, const int64_t* ids, int32_t* list, int32_t* n_out, llmrec_stream_t stream_) {
    LLMREC_CHECK_ARG(n >= 0 && n <= LLMREC_SORT_UNIQUE_MAX && n_out, "sort_unique_ids: 0 <= n <= %d", LLMREC_SORT_UNIQUE_MAX);
    LLMREC_CHECK_ARG(n == 0 || (ids && list), "sort_unique_ids: null pointer");
    int n_pow2 = 2;
    while (n_pow2 < n) n_pow2 <<= 1;
    if ((size_t)n_pow2 * sizeof(uint32_t) > 64 * 1024 && !big_lds_ok(0, reinterpret_cast<const void*>(sort_unique_block_kernel),
                                                                      LLMREC_SORT_UNIQUE_MAX * (int)sizeof(uint32_t))) {
        set_error("sort_unique_ids: %d ids need %zu B of LDS, which this device does not grant", (int)n, (size_t)n_pow2 * sizeof(uint32_t));
        return LLMREC_EUNSUPPORTED;
    }
    sort_unique_block_kernel<<<1, 1024, (size_t)n_pow2 * sizeof(uint32_t), (hipStream_t)stream_>>>((int)n, n_pow2, ids, list, n_out);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

int64_t llmrec_scatter_rows_workspace_bytes(int64_t n) {
    if (n < 0) return -1;
    return 2 * align_up(8 * n, 256) + align_up(n + (32ll << 20), 256);
}

int llmrec_scatter_rows_f32(int64_t n, const int64_t* ids, const float* rows, int64_t ldr, int32_t d, float alpha,
                            float* dst, int64_t ldd, void* workspace, int64_t workspace_bytes, llmrec_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    LLMREC_CHECK_ARG(n >= 0 && d > 0 && n < (1ll << 31), "scatter_rows: bad sizes");
    if (n == 0) return LLMREC_OK;
    LLMREC_CHECK_ARG(ids && rows && dst && ldr >= d && ldd >= d, "scatter_rows: null pointer or ld < d");
    if (!workspace || workspace_bytes < llmrec_scatter_rows_workspace_bytes(n)) {
        set_error("scatter_rows: workspace %lld < %lld", (long long)workspace_bytes, (long long)llmrec_scatter_rows_workspace_bytes(n));
        return LLMREC_EWORKSPACE;
    }
    char* ws = (char*)workspace;
    uint64_t* keys_a = (uint64_t*)ws;
    uint64_t* keys_b = (uint64_t*)(ws + align_up(8 * n, 256));
    void* temp = ws + 2 * align_up(8 * n, 256);
    size_t temp_avail = (size_t)align_up(n + (32ll << 20), 256), need = 0;
    if (n <= SCATTER_SORT_MAX) {                               // the per-step case: one block, keys sorted in LDS, no library call
        int n_pow2 = 2;
        while (n_pow2 < n) n_pow2 <<= 1;
        const size_t shmem = (size_t)n_pow2 * sizeof(uint64_t);
        // more than 64 KB of LDS needs the per-DEVICE attribute (a process-wide flag skipped the second device); a part that
        // does not grant it takes the radix-sort path below
        if (shmem <= 64 * 1024 || big_lds_ok(1, reinterpret_cast<const void*>(scatter_sort_block_kernel), SCATTER_SORT_MAX * (int)sizeof(uint64_t))) {
            scatter_sort_block_kernel<<<1, 1024, shmem, stream>>>((int)n, n_pow2, ids, keys_a);
            LLMREC_LAUNCH_CHECK();
            scatter_runs_kernel<<<(unsigned)ceil_div(n, 16), 256, 0, stream>>>(n, keys_a, rows, ldr, d, alpha, dst, ldd);
            LLMREC_LAUNCH_CHECK();
            return LLMREC_OK;
        }
    }
    scatter_keys_kernel<<<grid_for(n, 256), 256, 0, stream>>>(n, ids, keys_a);
    LLMREC_LAUNCH_CHECK();
    hipcub::DoubleBuffer<uint64_t> kbuf(keys_a, keys_b);
    LLMREC_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, need, kbuf, (int)n, 0, 64, stream));
    if (need > temp_avail) { set_error("scatter_rows: rocPRIM needs %zu B temp > %zu", need, temp_avail); return LLMREC_EWORKSPACE; }
    LLMREC_HIP(hipcub::DeviceRadixSort::SortKeys(temp, need, kbuf, (int)n, 0, 64, stream));
    scatter_runs_kernel<<<(unsigned)ceil_div(n, 16), 256, 0, stream>>>(n, kbuf.Current(), rows, ldr, d, alpha, dst, ldd);
    LLMREC_LAUNCH_CHECK();
    return LLMREC_OK;
}

}  // extern "C"

// sort.hip -- spatial ordering of large query batches (render.py:_sdf_at): Morton keys of the points' cells, a device radix
// sort of (key, index) pairs restricted to the key bits that carry locality, and the index-driven row gather / scatter.
// The sort itself is rocPRIM's device radix sort (a plain library sort, as rocBLAS is for a plain GEMM); what this file
// adds over torch.sort on the same keys: 32-bit indices instead of 64-bit (8 instead of 12 bytes per pair and pass), and only
// the bits [drop_bits, 30) of the key are sorted (3 passes instead of 4 when the low 6 bits -- the position inside a 4-cell
// block -- are left unsorted).  The order only schedules the work: results are written back through the same indices.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ia_common.h"

namespace {

constexpr int THREADS = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    return (v | (v << 2)) & 0x09249249u;
}

__global__ __launch_bounds__(THREADS) void morton_iota_kernel(int64_t n, const float* __restrict__ pts, float ox, float oy, float oz,
                                                               float inv_cell, uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const float fx = (pts[3 * i] - ox) * inv_cell, fy = (pts[3 * i + 1] - oy) * inv_cell, fz = (pts[3 * i + 2] - oz) * inv_cell;
    const uint32_t x = (uint32_t)fminf(fmaxf(fx, 0.0f), 1023.0f), y = (uint32_t)fminf(fmaxf(fy, 0.0f), 1023.0f),
                   z = (uint32_t)fminf(fmaxf(fz, 0.0f), 1023.0f);
    keys[i] = spread10(x) | (spread10(y) << 1) | (spread10(z) << 2);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(THREADS) void gather_rows3_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                    float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t j = order[i];
    dst[3 * i] = src[3 * j]; dst[3 * i + 1] = src[3 * j + 1]; dst[3 * i + 2] = src[3 * j + 2];
}

__global__ __launch_bounds__(THREADS) void scatter_f32_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                   float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    dst[order[i]] = src[i];
}

__global__ __launch_bounds__(THREADS) void scatter_rows3_i32_kernel(int64_t n, const float* __restrict__ src, const int32_t* __restrict__ order,
                                                                     float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t j = order[i];
    dst[3 * j] = src[3 * i]; dst[3 * j + 1] = src[3 * i + 1]; dst[3 * j + 2] = src[3 * i + 2];
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// (measured: a custom onesweep config with 10 instead of rocPRIM's tuned 8 bits per pass -- 3 passes over the 30 key bits instead of 4 --
//  is no faster: 10.9 against 10.6 ms per headline step for the eight sorts)
size_t sort_storage_bytes(int64_t n, int drop_bits)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config, uint32_t*, uint32_t*, uint32_t*, uint32_t*>(
        nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, (unsigned)drop_bits, 30u, (hipStream_t)0, false);
    return bytes;
}

}  // namespace

IA_EXPORT size_t ia_morton_order_tmp_bytes(int64_t n)
{
    if (n <= 0) return 256;
    return 3 * align256((size_t)n * 4) + align256(sort_storage_bytes(n, 0)) + 256;
}

// order [n] int32: the permutation that lists the points by the Morton code of their cell (cell size 1 / inv_cell, 10 bits per
// axis from `origin`), stable; drop_bits (0, 3 or 6) low key bits are ignored by the sort.
IA_EXPORT int ia_morton_order(int64_t n, const float* pts, const float* origin_host3, float inv_cell, int drop_bits, int32_t* order,
                              void* tmp, size_t tmp_bytes, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    IA_REQUIRE(n < ((int64_t)1 << 31), "ia_morton_order: n must stay below 2^31");
    IA_REQUIRE(drop_bits >= 0 && drop_bits < 30, "ia_morton_order: drop_bits out of range");
    IA_REQUIRE(tmp != nullptr && tmp_bytes >= ia_morton_order_tmp_bytes(n), "ia_morton_order: tmp too small (ia_morton_order_tmp_bytes)");
    IA_REQUIRE((reinterpret_cast<uintptr_t>(tmp) & 255) == 0, "ia_morton_order: tmp must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char* p = reinterpret_cast<char*>(tmp);
    const size_t col = align256((size_t)n * 4);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(p);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(p + col);
    uint32_t* idx_in = reinterpret_cast<uint32_t*>(p + 2 * col);
    void* storage = p + 3 * col;
    size_t storage_bytes = sort_storage_bytes(n, drop_bits);
    morton_iota_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, s>>>(n, pts, origin_host3[0], origin_host3[1], origin_host3[2], inv_cell, keys_in,
                                                                 idx_in);
    const hipError_t e = rocprim::radix_sort_pairs(storage, storage_bytes, keys_in, keys_out, idx_in, reinterpret_cast<uint32_t*>(order),
                                                   (size_t)n, (unsigned)drop_bits, 30u, s, false);
    if (e != hipSuccess) {
        ia::set_error("ia_morton_order: rocprim::radix_sort_pairs: %s", hipGetErrorString(e));
        return IA_ERR_LAUNCH;
    }
    return ia::check_launch("ia_morton_order");
}

IA_EXPORT int ia_gather_rows3_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    gather_rows3_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_gather_rows3_i32");
}

IA_EXPORT int ia_scatter_f32_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    scatter_f32_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_scatter_f32_i32");
}

IA_EXPORT int ia_scatter_rows3_i32(int64_t n, const float* src, const int32_t* order, float* dst, ia_stream_t stream)
{
    if (n == 0) return IA_OK;
    scatter_rows3_i32_kernel<<<ia::cdiv(n, THREADS), THREADS, 0, (hipStream_t)stream>>>(n, src, order, dst);
    return ia::check_launch("ia_scatter_rows3_i32");
}

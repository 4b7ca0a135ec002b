// core.hip -- error reporting, version, exclusive prefix sums (plumbing for the
// two-phase "count -> allocate -> fill" protocol of the data-dependent operators).
#include <stdarg.h>
#include <string.h>

#include "ia_common.h"

namespace ia {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ia

IA_EXPORT int ia_version(void) { return 100; }
IA_EXPORT const char* ia_last_error(void) { return ia::g_err; }

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// wave64 inclusive scan by shuffles
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// one tile per block: out = exclusive scan within the tile, tile_sums[tile] = tile total
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_kernel(const T* in, T* out, T* tile_sums, int64_t n)
{
    __shared__ T wave_tot[SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)tid * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T local = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        v[k] = (base + k < n) ? in[base + k] : (T)0;
        local += v[k];
    }
    T inc = wave_inclusive_scan(local, lane);
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    T wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) {
        T t = wave_tot[w];
        if (w < wid) wave_off += t;
        total += t;
    }
    T run = wave_off + inc - local;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (tid == 0) tile_sums[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(T* __restrict__ out, const T* __restrict__ tile_offs,
                                                                 int64_t n)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    const T off = tile_offs[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++)
        if (base + k < n) out[base + k] += off;
}

template <typename T>
__global__ void copy_one_kernel(const T* src, T* dst) { *dst = *src; }
template <typename T>
__global__ void zero_one_kernel(T* dst) { *dst = 0; }

// Small inputs (the per-ray count tables of a 4096-ray training batch, the tile sums of a larger scan): ONE workgroup walks the whole array
// in chunks of 1024 x SCAN_ITEMS elements with a running carry and writes the total itself -- one launch instead of the four of the tiled
// protocol (tiles, tile sums, copy of the total, add-back); the 4096-ray step issued 75 scan launches (profiles/r05_config4_kernel_stats.csv).
constexpr int SCAN_SMALL_THREADS = 1024;
constexpr int64_t SCAN_SMALL_MAX = 1 << 15;      // 8 chunks of 4096: above that the tiled protocol's parallel tiles win
template <typename T>
__global__ __launch_bounds__(SCAN_SMALL_THREADS) void scan_small_kernel(const T* in, T* out, T* total, int64_t n)
{
    __shared__ T wave_tot[SCAN_SMALL_THREADS / 64];
    __shared__ T carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t chunk = 0; chunk < n; chunk += (int64_t)SCAN_SMALL_THREADS * SCAN_ITEMS) {
        const int64_t base = chunk + (int64_t)tid * SCAN_ITEMS;
        T v[SCAN_ITEMS];
        T local = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            v[k] = (base + k < n) ? in[base + k] : (T)0;
            local += v[k];
        }
        const T inc = wave_inclusive_scan(local, lane);
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        T wave_off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SCAN_SMALL_THREADS / 64; w++) {
            const T t = wave_tot[w];
            if (w < wid) wave_off += t;
            tot += t;
        }
        T run = carry_s + wave_off + inc - local;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; k++) {
            if (base + k < n) out[base + k] = run;
            run += v[k];
        }
        __syncthreads();                      // every thread has read carry_s and wave_tot
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry_s;
}

template <typename T>
int scan_impl(const T* in, T* out, T* total, int64_t n, void* tmp, hipStream_t s)
{
    if (n <= 0) {
        if (total) zero_one_kernel<T><<<1, 1, 0, s>>>(total);
        return ia::check_launch("scan(empty)");
    }
    if (n <= SCAN_SMALL_MAX) {
        scan_small_kernel<T><<<1, SCAN_SMALL_THREADS, 0, s>>>(in, out, total, n);
        return ia::check_launch("scan(small)");
    }
    // level buffers carved from tmp (8-byte slots)
    int64_t* slots = (int64_t*)tmp;
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    T* sums = (T*)slots;
    scan_tiles_kernel<T><<<(int)tiles, SCAN_THREADS, 0, s>>>(in, out, sums, n);
    if (tiles == 1) {
        if (total) copy_one_kernel<T><<<1, 1, 0, s>>>(sums, total);
        return ia::check_launch("scan");
    }
    // scan the tile sums in place (recursively), then add back
    int r = scan_impl<T>(sums, sums, total, tiles, (void*)(slots + tiles), s);
    if (r != IA_OK) return r;
    scan_add_kernel<T><<<(int)tiles, SCAN_THREADS, 0, s>>>(out, sums, n);
    return ia::check_launch("scan");
}

}  // namespace

IA_EXPORT int64_t ia_scan_tmp_bytes(int64_t n)
{
    int64_t slots = 16;
    while (n > 1) {
        n = (n + SCAN_TILE - 1) / SCAN_TILE;
        slots += n + 2;
    }
    return slots * 8;
}

IA_EXPORT int ia_exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t* total, int64_t n, void* tmp,
                                    ia_stream_t stream)
{
    return scan_impl<int64_t>(in, out, total, n, tmp, (hipStream_t)stream);
}

IA_EXPORT int ia_exclusive_scan_i32(const int32_t* in, int32_t* out, int32_t* total, int64_t n, void* tmp,
                                    ia_stream_t stream)
{
    return scan_impl<int32_t>(in, out, total, n, tmp, (hipStream_t)stream);
}

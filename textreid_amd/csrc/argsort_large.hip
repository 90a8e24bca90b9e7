// Full descending argsort of similarity rows that do not fit the in-LDS bitonic sort of retrieval.hip (G > 16384:
// rank(get_mAP=True) on ICFG-PEDES-size galleries, evaluation.py:14).  Keys are packed as
// (order-reversed float bits << 32 | column) and sorted per row by rocPRIM's segmented radix sort on the high 32 bits
// only - the sort is stable and the packed input is in column order, so ties come out lowest-column-first exactly
// like the bitonic kernel.  All temporary storage is the caller's workspace (the library never allocates).

#include <hipcub/hipcub.hpp>

#include "split_common.h"

namespace trid {

__global__ __launch_bounds__(256) void argsort_pack_kernel(const float* __restrict__ sim, int ld, int G, long long total,
                                                           unsigned long long* __restrict__ keys) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long q = i / G;
        const int c = (int)(i - q * G);
        float v = sim[q * ld + c];
        if (v == 0.f) v = 0.f;  // -0 and +0 tie, as in a float comparison
        unsigned b = __builtin_bit_cast(unsigned, v);
        b ^= (b >> 31) ? 0xffffffffu : 0x80000000u;  // ascending unsigned order == ascending float order
        keys[i] = ((unsigned long long)(~b) << 32) | (unsigned)c;  // ascending key == descending value
    }
}

__global__ __launch_bounds__(256) void argsort_unpack_kernel(const unsigned long long* __restrict__ keys, long long total,
                                                             long long* __restrict__ out_idx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        out_idx[i] = (long long)(keys[i] & 0xffffffffull);
}

struct RowOffset {
    int G;
    __host__ __device__ __forceinline__ int operator()(int q) const { return q * G; }
};
typedef hipcub::TransformInputIterator<int, RowOffset, hipcub::CountingInputIterator<int>> OffsetIt;

static hipError_t sort_rows(void* temp, size_t& temp_bytes, const unsigned long long* in, unsigned long long* out, int Q, int G,
                            hipStream_t stream) {
    hipcub::CountingInputIterator<int> count(0);
    OffsetIt begin(count, RowOffset{G});
    OffsetIt end(count + 1, RowOffset{G});
    return hipcub::DeviceSegmentedRadixSort::SortKeys(temp, temp_bytes, in, out, Q * G, Q, begin, end, 32, 64, stream);
}

int argsort_rows_desc_large(const float* sim, int ld, int Q, int G, int64_t* out_idx, void* ws, long long ws_bytes, void* stream_);

}  // namespace trid

using namespace trid;

extern "C" long long trid_argsort_ws_bytes(int Q, int G) {
    if (Q <= 0 || G <= 16384) return 0;  // the in-LDS path needs no workspace
    if ((long long)Q * G >= (1ll << 31)) return 0;
    size_t temp = 0;
    if (sort_rows(nullptr, temp, nullptr, nullptr, Q, G, nullptr) != hipSuccess) return 0;
    return 2LL * Q * G * 8 + (long long)((temp + 255) / 256 * 256) + 256;
}

// called by trid_argsort_rows_desc_f32 (retrieval.hip) for G > 16384
int trid::argsort_rows_desc_large(const float* sim, int ld, int Q, int G, int64_t* out_idx, void* ws, long long ws_bytes,
                                  void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    TRID_REQUIRE((long long)Q * G < (1ll << 31), "trid_argsort_rows_desc_f32: Q * G = %lld exceeds 2^31 - sort the rows in batches",
                 (long long)Q * G);
    const long long need = trid_argsort_ws_bytes(Q, G);
    TRID_REQUIRE(ws != nullptr && need > 0 && ws_bytes >= need,
                 "trid_argsort_rows_desc_f32: a gallery of %d rows needs a workspace of trid_argsort_ws_bytes(Q, G) = %lld bytes", G, need);
    TRID_REQUIRE(((uintptr_t)ws & 255) == 0, "trid_argsort_rows_desc_f32: workspace must be 256-byte aligned");
    const long long total = (long long)Q * G;
    unsigned long long* k0 = reinterpret_cast<unsigned long long*>(ws);
    unsigned long long* k1 = k0 + total;
    void* temp = k1 + total;
    size_t temp_bytes = (size_t)(need - 2 * total * 8);
    hipLaunchKernelGGL(argsort_pack_kernel, dim3(grid_for(total, 256 * 4)), dim3(256), 0, stream, sim, ld, G, total, k0);
    const hipError_t e = sort_rows(temp, temp_bytes, k0, k1, Q, G, stream);
    if (e != hipSuccess) {
        set_error("trid_argsort_rows_desc_f32: segmented radix sort failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    hipLaunchKernelGGL(argsort_unpack_kernel, dim3(grid_for(total, 256 * 4)), dim3(256), 0, stream, k1, total, (long long*)out_idx);
    return check_launch("trid_argsort_rows_desc_f32");
}

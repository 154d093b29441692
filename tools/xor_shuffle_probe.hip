#include <hip/hip_runtime.h>
template <int OFF>
__device__ __forceinline__ int xor_i(int v) {
    if constexpr (OFF == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    else if constexpr (OFF == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    else if constexpr (OFF == 4) {
        const int a = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0xf, false); // row_shl:4 : lane i <- i + 4
        const int b = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); // row_shr:4 : lane i <- i - 4
        return (threadIdx.x & 4) ? b : a;
    } else if constexpr (OFF == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false); // row_ror:8
    else if constexpr (OFF == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
__global__ void k(int* out) {
    int v = threadIdx.x * 3 + 7;
    out[threadIdx.x + 0 * 64] = xor_i<1>(v);
    out[threadIdx.x + 1 * 64] = xor_i<2>(v);
    out[threadIdx.x + 2 * 64] = xor_i<4>(v);
    out[threadIdx.x + 3 * 64] = xor_i<8>(v);
    out[threadIdx.x + 4 * 64] = xor_i<16>(v);
    out[threadIdx.x + 5 * 64] = xor_i<32>(v);
}
int main() {
    int* d; hipMalloc(&d, 6 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int offs[6] = {1, 2, 4, 8, 16, 32}; int bad = 0;
    for (int t = 0; t < 6; ++t) for (int l = 0; l < 64; ++l) if (h[t * 64 + l] != (l ^ offs[t]) * 3 + 7) { if (bad < 10) printf("off %d lane %d got %d want %d\n", offs[t], l, h[t*64+l], (l ^ offs[t]) * 3 + 7); ++bad; }
    printf("bad %d\n", bad);
    return bad != 0;
}

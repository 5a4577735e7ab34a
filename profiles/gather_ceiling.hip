// random 64-byte-sector gather ceiling: every lane reads one dword from a pseudo-random 64 B sector of a big table
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(256) k_gather(const uint32_t* tab, uint64_t n_sectors, uint32_t iters, uint32_t* out, int dep) {
    uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    if (dep == 0) {
        for (uint32_t i = 0; i < iters; i += 8) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                x = x * 6364136223846793005ull + 1442695040888963407ull;
                uint64_t s = (uint64_t)(((unsigned __int128)(x >> 11 << 11) * n_sectors) >> 64);
                v[j] = __builtin_nontemporal_load(tab + s * 16 + (x & 15));
            }
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
    } else {                                                    // dependent chain: next address depends on the loaded value (like the walk)
        for (uint32_t i = 0; i < iters; i++) {
            x = x * 6364136223846793005ull + 1442695040888963407ull + acc;
            uint64_t s = (uint64_t)(((unsigned __int128)(x >> 11 << 11) * n_sectors) >> 64);
            acc += tab[s * 16 + (x & 15)];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
extern "C" int run_gather(uint64_t table_bytes, uint32_t blocks, uint32_t iters, int dep, float* ms_out) {
    uint32_t *tab, *out;
    if (hipMalloc(&tab, table_bytes) != hipSuccess) return -1;
    if (hipMalloc(&out, blocks * 256 * 4) != hipSuccess) return -1;
    hipMemset(tab, 1, table_bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, table_bytes / 64, iters / 4, out, dep);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, table_bytes / 64, iters, out, dep);
    hipEventRecord(b, 0);
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    hipEventElapsedTime(ms_out, a, b);
    hipFree(tab); hipFree(out); return 0;
}

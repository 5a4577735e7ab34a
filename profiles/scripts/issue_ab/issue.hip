// What one wave alone on its SIMD pays per instruction (gfx950): dependent and independent chains of the instruction kinds the
// range coder's serial step is made of.  One workgroup of 64 threads per CU; cycles from s_memtime (100 MHz-independent core clock
// counter: reported beside the wall clock).  hipcc --offload-arch=gfx950 -O2 issue.hip -o issue && ./issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND> __global__ void __launch_bounds__(64) k(uint64_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d = seed + 7;
    uint64_t A = ((uint64_t)a << 32) | b, B = ((uint64_t)c << 32) | d;
    uint32_t sa = seed, sb = seed * 5 + 3;
    asm volatile("" : "+s"(sa), "+s"(sb));
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 1) { REP64(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (KIND == 2) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 3) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));) }
        if (KIND == 4) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 5) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(A) : "v"(a), "v"(b) : "vcc");) }
        if (KIND == 6) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(A), "+v"(B) : "v"(a), "v"(b) : "vcc");) }
        if (KIND == 7) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(A) : "v"(B));) }
        if (KIND == 8) { REP64(asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(A));) }
        if (KIND == 9) { REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(sa) : "s"(sb) : "scc");) }
        if (KIND == 10) { REP64(asm volatile("s_mul_i32 %0, %0, %1" : "+s"(sa) : "s"(sb));) }
        if (KIND == 11) { REP64(asm volatile("s_mul_hi_u32 %0, %0, %1" : "+s"(sa) : "s"(sb));) }
        if (KIND == 12) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 13) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(A) : "v"(B));) }
        if (KIND == 14) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );) }
        if (KIND == 15) { REP64(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        if (KIND == 16) { REP64(asm volatile("v_add_u32 %0, %0, %2\n\ts_add_u32 %1, %1, %3" : "+v"(a), "+s"(sa) : "v"(b), "s"(sb) : "scc");) }
        if (KIND == 17) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = a + b + c + d + (uint32_t)A + (uint32_t)B + sa;
}
template <int KIND> void run(const char* name, int per_rep, uint64_t* d_out, int blocks) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * 2);
    hipMemcpy(h.data(), d_out, blocks * 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64 * per_rep;
    printf("%-46s %7.2f counter ticks / instr   %7.2f ns / instr (wall, one wave per CU)\n", name, (double)h[0] / n, ms * 1e6 / n);
}
int main() {
    uint64_t* d; hipMalloc(&d, 4096 * 16);
    const int B = 256;
    run<0>("v_add_u32, dependent", 1, d, B);
    run<1>("v_add_u32, 4 independent", 4, d, B);
    run<2>("v_mul_lo_u32, dependent", 1, d, B);
    run<3>("v_mul_lo_u32, 4 independent", 4, d, B);
    run<4>("v_mul_hi_u32, dependent", 1, d, B);
    run<5>("v_mad_u64_u32, dependent (accumulator)", 1, d, B);
    run<6>("v_mad_u64_u32, 2 independent", 2, d, B);
    run<7>("v_lshl_add_u64, dependent", 1, d, B);
    run<8>("v_lshlrev_b64, dependent", 1, d, B);
    run<9>("s_add_u32, dependent", 1, d, B);
    run<10>("s_mul_i32, dependent", 1, d, B);
    run<11>("s_mul_hi_u32, dependent", 1, d, B);
    run<12>("v_mul_u32_u24, dependent", 1, d, B);
    run<13>("v_fma_f64, dependent", 1, d, B);
    run<14>("v_cndmask_b32, dependent", 1, d, B);
    run<15>("v_mov_b32_dpp row_shr:1, dependent", 1, d, B);
    run<16>("v_add_u32 + s_add_u32 pairs (both dependent)", 2, d, B);
    run<17>("v_mul_lo_u32 + 3 dependent v_add (other reg)", 4, d, B);
    return 0;
}

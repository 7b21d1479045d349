// Calibration: issue rate of the 32x32-bit integer multiplies Philox needs, against v_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O2 imul_rate.hip -o imul_rate; run: imul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 4000;
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned m) {
    unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long y0 = x0, y1 = x1, y2 = x2, y3 = x3;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %6, %5, 0\n v_mad_u64_u32 %2, vcc, %7, %5, 0\n v_mad_u64_u32 %3, vcc, %8, %5, 0\n"
                                       : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(x0), "v"(m), "v"(x1), "v"(x2), "v"(x3) : "vcc");
            if (MODE == 1) asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 2) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 3) asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 4) asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(y0 + y1 + y2 + y3);
}
template <int MODE>
static void run(const char* name, unsigned* out) {
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 0xD2511F53u);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 0xD2511F53u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITER * 64;
    printf("%-16s %.3f ms: %.2f clocks per wave instruction and SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / insts);
}
int main() {
    unsigned* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(unsigned));
    run<0>("v_mad_u64_u32", out); run<1>("v_mul_hi_u32", out); run<2>("v_mul_lo_u32", out); run<3>("v_mul_u32_u24", out); run<4>("v_xor_b32", out);
    return 0;
}

// Calibration: a kernel with a known number of VALU instructions per wave — ITER iterations x 64 v_fma_f32 in
// 8 independent chains — to check what SQ_INSTS_VALU counts (wave-level instructions: exact) and to measure
// the issue rate of plain (un-packed) f32 code. Build: hipcc --offload-arch=gfx950 -O2 valu_count.hip -o valu_count;
// run: valu_count <waves per SIMD>. MI355X: 57.7 / 101.9 / 111.8 TFLOP/s at 1 / 4 / 8 waves per SIMD, i.e. a
// wave64 v_fma_f32 every 2.8 cycles per SIMD at the nominal 2.4 GHz (the clock under this load read 1.97 GHz
// from GRBM_GUI_ACTIVE: 2.5 cycles) — faster than the 4 cycles of a 16-lane pipe, slower than the 2 of the
// packed peak.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 20000;
__global__ __launch_bounds__(256) void fma_kernel(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main(int argc, char** argv) {
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd x 4 SIMDs / 4 waves per block)
    float* out; (void)hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    fma_kernel<<<blocks, 256>>>(out, 1.0001f, 0.5f);
    (void)hipEventRecord(e0);
    fma_kernel<<<blocks, 256>>>(out, 1.0001f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITER * 64;
    printf("waves/SIMD %d: %.3f ms, %.3e wave-level v_fma per launch, %.2f TFLOP/s, cycles per instruction per SIMD at 2.4 GHz: %.2f\n",
           waves_per_simd, ms, insts, insts * 64 * 2 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 * 1024 / insts);
    return 0;
}

// Calibration: issue rate of the VALU instruction classes the render kernels are made of (8 waves per SIMD, four
// independent chains per wave). Build: hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates; run: valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2000;
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY(TXT, ...) asm volatile(TXT : __VA_ARGS__)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float m) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float y = m * 3.f, z0 = x0 * 2, z1 = x1 * 2, z2 = x2 * 2, z3 = x3 * 2;
    asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "vcc");
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x1, x2}, p3 = {x3, x0}, pm = {m, m};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) BODY("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 1) BODY("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 2) BODY("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 3) BODY("v_max3_f32 %0, %0, %4, %1\n v_max3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_max3_f32 %3, %3, %4, %0\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 4) BODY("v_cmp_le_f32 vcc, %0, %4\n v_cmp_le_f32 vcc, %1, %4\n v_cmp_le_f32 vcc, %2, %4\n v_cmp_le_f32 vcc, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 5) BODY("v_cmp_le_f32 s[20:21], %0, %4\n v_cmp_le_f32 s[22:23], %1, %4\n v_cmp_le_f32 s[24:25], %2, %4\n v_cmp_le_f32 s[26:27], %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (MODE == 6) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 7) BODY("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n", "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            if (MODE == 8) BODY("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n", "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            if (MODE == 9) BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 10) BODY("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 11) BODY("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 12) BODY("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 14) BODY("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "s20", "s21");
            if (MODE == 15) BODY("v_cndmask_b32 %0, %5, %4, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n", "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"(m), "v"(y) : "vcc");
            if (MODE == 16) BODY("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n v_fma_f32 %5, %5, %4, %4\n v_fma_f32 %6, %6, %4, %4\n v_fma_f32 %7, %7, %4, %4\n v_fma_f32 %8, %8, %4, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m), "v"(z0), "v"(z1), "v"(z2), "v"(z3));
            if (MODE == 17) BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 18) BODY("v_mul_f32_e64 %0, %0, %4\n v_mul_f32_e64 %1, %1, %4\n v_mul_f32_e64 %2, %2, %4\n v_mul_f32_e64 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            if (MODE == 19) BODY("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 20) BODY("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_le_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 21) BODY("v_cmp_le_f32 s[20:21], %0, %4\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cmp_le_f32 s[22:23], %2, %4\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "s20", "s21", "s22", "s23");
            if (MODE == 13) BODY("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %0\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %2\n", "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = z0 + z1 + z2 + z3 + y + x0 + x1 + x2 + x3 + p0.x + p1.y + p2.x + p3.y;
}
template <int MODE>
static void run(const char* name, float* out) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITER * 64;
    printf("%-22s %.2f clocks per wave instruction and SIMD at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 * 1024 / insts);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", out); run<1>("v_mul_f32", out); run<2>("v_min_f32", out); run<3>("v_max3_f32", out); run<4>("v_cmp_le_f32 vcc", out);
    run<5>("v_cmp_le_f32 sgpr", out); run<6>("v_cndmask_b32", out); run<7>("v_pk_fma_f32", out); run<8>("v_pk_mul_f32", out);
    run<9>("v_rcp_f32", out); run<10>("v_rsq_f32", out); run<11>("v_sin_f32", out); run<12>("v_cvt_f32_u32", out); run<13>("v_lshl_add_u64", out);
    run<14>("v_cndmask_b32 sgpr", out); run<15>("v_cndmask_b32 indep", out); run<16>("v_fma_f32 x8 (8 instr)", out); run<17>("v_add_f32", out); run<18>("v_mul_f32_e64", out);
    run<19>("v_cndmask_b32_e64 vcc", out); run<20>("v_cmp + v_cndmask vcc", out); run<21>("v_cmp + v_cndmask sgpr", out);
    return 0;
}

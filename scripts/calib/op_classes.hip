// Calibration: vector-unit time of the instruction kinds the render kernels are made of, beyond the floating-point ones
// of valu_rates.hip (8 waves per SIMD, four independent chains per wave; clocks per wave instruction and SIMD).
// Build: hipcc --offload-arch=gfx950 -O2 op_classes.hip -o op_classes
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2000;
typedef float f2 __attribute__((ext_vector_type(2)));
#define FOUR(OP) OP " %0, %0, %4\n " OP " %1, %1, %4\n " OP " %2, %2, %4\n " OP " %3, %3, %4\n"
#define FOUR3(OP) OP " %0, %0, %4, %1\n " OP " %1, %1, %4, %2\n " OP " %2, %2, %4, %3\n " OP " %3, %3, %4, %0\n"
#define FOUR1(OP) OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3\n"
#define RUN4(TXT) asm volatile(TXT : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m))
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t m) {
    uint32_t x0 = threadIdx.x + 3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long q0 = x0, q1 = x1, q2 = x2, q3 = x3;
    f2 p0 = {1.f + x0, 2.f}, p1 = {2.f + x0, 3.f}, p2 = {3.f + x0, 1.f}, p3 = {4.f + x0, 2.f}, pm = {1.0001f, 1.0001f};
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) RUN4(FOUR("v_and_b32"));
            if (MODE == 1) RUN4(FOUR("v_or_b32"));
            if (MODE == 2) RUN4(FOUR("v_xor_b32"));
            if (MODE == 3) RUN4(FOUR("v_add_u32"));
            if (MODE == 4) RUN4(FOUR("v_lshlrev_b32"));
            if (MODE == 5) RUN4(FOUR("v_sub_f32"));
            if (MODE == 6) RUN4(FOUR("v_fmac_f32"));
            if (MODE == 7) RUN4(FOUR3("v_fma_f32"));
            if (MODE == 8) RUN4(FOUR3("v_med3_f32"));
            if (MODE == 9) RUN4(FOUR3("v_bfe_u32"));
            if (MODE == 10) RUN4(FOUR3("v_and_or_b32"));
            if (MODE == 11) RUN4(FOUR3("v_add3_u32"));
            if (MODE == 12) RUN4(FOUR3("v_perm_b32"));
            if (MODE == 13) RUN4(FOUR("v_mul_lo_u32"));
            if (MODE == 14) RUN4(FOUR("v_mul_hi_u32"));
            if (MODE == 15) RUN4(FOUR1("v_mov_b32"));
            if (MODE == 16) RUN4(FOUR1("v_floor_f32"));
            if (MODE == 17) RUN4(FOUR1("v_fract_f32"));
            if (MODE == 18) RUN4(FOUR1("v_cvt_f32_ubyte0"));
            if (MODE == 19) RUN4(FOUR("v_max_f32"));
            if (MODE == 20) RUN4(FOUR("v_max_u32"));
            if (MODE == 21) RUN4(FOUR("v_mul_f32"));
            if (MODE == 22) RUN4("v_bitop3_b32 %0, %0, %4, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %0 bitop3:0x96\n");
            if (MODE == 23) asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m), "v"(x0) : "vcc");
            if (MODE == 24) asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            if (MODE == 25) asm volatile("v_pk_mov_b32 %0, %0, %4 op_sel:[1,0]\n v_pk_mov_b32 %1, %1, %4 op_sel:[1,0]\n v_pk_mov_b32 %2, %2, %4 op_sel:[1,0]\n v_pk_mov_b32 %3, %3, %4 op_sel:[1,0]\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
            if (MODE == 26) RUN4("v_max_f32 %0, |%0|, %4\n v_max_f32 %1, |%1|, %4\n v_max_f32 %2, |%2|, %4\n v_max_f32 %3, |%3|, %4\n");
            if (MODE == 27) RUN4("v_add_f32 %0, |%0|, %4\n v_add_f32 %1, |%1|, %4\n v_add_f32 %2, |%2|, %4\n v_add_f32 %3, |%3|, %4\n");
            if (MODE == 28) RUN4("v_mul_f32 %0, 0x3f8ccccd, %0\n v_mul_f32 %1, 0x3f8ccccd, %1\n v_mul_f32 %2, 0x3f8ccccd, %2\n v_mul_f32 %3, 0x3f8ccccd, %3\n");
            if (MODE == 29) RUN4(FOUR("v_ashrrev_i32"));
            if (MODE == 30) RUN4(FOUR("v_min_u32"));
            if (MODE == 31) RUN4(FOUR("v_ldexp_f32"));
            if (MODE == 32) RUN4(FOUR3("v_lshl_or_b32"));
            if (MODE == 33) RUN4(FOUR3("v_lshl_add_u32"));
            if (MODE == 34) RUN4(FOUR3("v_min3_f32"));
            if (MODE == 35) RUN4(FOUR("v_sub_u32"));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)(p0.x + p1.y + p2.x + p3.y);
}
template <int MODE>
static void run(const char* name, uint32_t* out) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 3u);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 3u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITER * 64;
    printf("%-22s %.2f\n", name, ms * 1e-3 * 2.4e9 * 1024 / insts);
    fflush(stdout);
}
int main() {
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t));
    printf("clocks per wave instruction and SIMD at 2.4 GHz\n");
    run<0>("v_and_b32", out); run<1>("v_or_b32", out); run<2>("v_xor_b32", out); run<3>("v_add_u32", out); run<35>("v_sub_u32", out); run<4>("v_lshlrev_b32", out);
    run<29>("v_ashrrev_i32", out); run<5>("v_sub_f32", out); run<21>("v_mul_f32", out); run<28>("v_mul_f32 literal", out); run<27>("v_add_f32 |x|", out);
    run<6>("v_fmac_f32", out); run<7>("v_fma_f32", out); run<19>("v_max_f32", out); run<26>("v_max_f32 |x|", out); run<20>("v_max_u32", out); run<30>("v_min_u32", out);
    run<8>("v_med3_f32", out); run<34>("v_min3_f32", out); run<9>("v_bfe_u32", out); run<10>("v_and_or_b32", out); run<32>("v_lshl_or_b32", out); run<33>("v_lshl_add_u32", out);
    run<11>("v_add3_u32", out); run<12>("v_perm_b32", out); run<22>("v_bitop3_b32", out); run<13>("v_mul_lo_u32", out); run<14>("v_mul_hi_u32", out); run<23>("v_mad_u64_u32", out);
    run<15>("v_mov_b32", out); run<16>("v_floor_f32", out); run<17>("v_fract_f32", out); run<18>("v_cvt_f32_ubyte0", out); run<31>("v_ldexp_f32", out);
    run<24>("v_pk_add_f32", out); run<25>("v_pk_mov_b32", out);
    return 0;
}

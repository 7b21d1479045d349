// Calibration: what a select costs by where its condition comes from. v_cndmask_b32 in its 32-bit encoding reads VCC
// implicitly, in its 64-bit encoding any scalar pair; the condition is written by a vector compare or by scalar code
// (s_and_b64 of several compares, s_mov). 8 waves per SIMD, four independent chains per wave.
// Build: hipcc --offload-arch=gfx950 -O2 cndmask_rates.hip -o cndmask_rates
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITER = 2000;
#define BODY(TXT, ...) asm volatile(TXT : __VA_ARGS__)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float m) {
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 s[22:23], 0x3333\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "s22", "s23", "vcc");
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // 0: e32, VCC written once by scalar code before the loop
            if (MODE == 0) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 1: scalar write of VCC, then four e32 selects
            if (MODE == 1) BODY("s_and_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc", "scc");
            // 2: vector compare into VCC, then four e32 selects
            if (MODE == 2) BODY("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 3: scalar write of a pair, then four e64 selects on it
            if (MODE == 3) BODY("s_and_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %0, %0, %4, s[24:25]\n v_cndmask_b32_e64 %1, %1, %4, s[24:25]\n v_cndmask_b32_e64 %2, %2, %4, s[24:25]\n v_cndmask_b32_e64 %3, %3, %4, s[24:25]\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "s24", "s25", "scc");
            // 4: one e32 select on a VCC written by scalar code long ago among seven adds
            if (MODE == 4) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 5: eight adds (the baseline of 4)
            if (MODE == 5) BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m));
            // 6: compare into a pair, scalar and with another pair into VCC, one e32 select (the compiled range test)
            if (MODE == 6) BODY("v_cmp_le_f32 s[24:25], %0, %4\n s_and_b64 vcc, s[24:25], s[20:21]\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %2, %2, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc", "s24", "s25", "scc");
            // 7: the same with the result in a pair and the e64 select
            if (MODE == 7) BODY("v_cmp_le_f32 s[24:25], %0, %4\n s_and_b64 s[26:27], s[24:25], s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[26:27]\n v_add_f32 %2, %2, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "s24", "s25", "s26", "s27", "scc");
            // 8: e32 selects with a literal-free different source order (src0 constant register)
            if (MODE == 8) BODY("v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 9-11: e32 selects on one VCC with one, two, three adds between them
            if (MODE == 9) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %2, %2, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 10) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            if (MODE == 11) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_add_f32 %1, %1, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 12: an e32 select next to an e64 select on a pair
            if (MODE == 12) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 13: e64 selects that name VCC
            if (MODE == 13) BODY("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 14: two e32 selects of one condition, then two adds (the compiled select of a vector's components)
            if (MODE == 14) BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
            // 15: carry chain through VCC (64-bit add), e32
            if (MODE == 15) BODY("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n", "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(m) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE>
static void run(const char* name, float* out, int per_group) {
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0001f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double groups = (double)blocks * 4 * ITER * 16;
    fflush(stdout);
    printf("%-58s %.2f clocks per group of %d vector instructions and SIMD at 2.4 GHz\n", name, ms * 1e-3 * 2.4e9 * 1024 / groups, per_group);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("4 x cndmask e32, VCC set by scalar code before the loop", out, 4);
    run<1>("s_and -> VCC, 4 x cndmask e32", out, 4);
    run<2>("v_cmp -> VCC, 4 x cndmask e32", out, 5);
    run<3>("s_and -> pair, 4 x cndmask e64", out, 4);
    run<4>("1 x cndmask e32 (old scalar VCC) + 7 x add", out, 8);
    run<5>("8 x add", out, 8);
    run<6>("v_cmp -> pair, s_and -> VCC, cndmask e32, add", out, 3);
    run<7>("v_cmp -> pair, s_and -> pair, cndmask e64, add", out, 3);
    run<8>("v_cmp -> VCC, cndmask e32, 2 x add", out, 4);
    run<9>("cndmask e32, add, cndmask e32, add", out, 4);
    run<10>("cndmask e32, 2 x add, cndmask e32, 2 x add", out, 6);
    run<11>("cndmask e32, 3 x add", out, 4);
    run<12>("cndmask e32, cndmask e64 pair, twice", out, 4);
    run<13>("4 x cndmask e64 naming VCC", out, 4);
    run<14>("2 x cndmask e32, 2 x add", out, 4);
    run<15>("add_co, addc_co, twice (carry through VCC)", out, 4);
    return 0;
}

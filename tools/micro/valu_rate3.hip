// VALU issue-rate microbenchmark, part 2 (select/compare forms and integer ops); same method as valu_rate2.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, const float* in, int iters, float s0) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], a2 = a0 + 1, a3 = a1 + 1, x = a0 + 2;
    asm volatile("s_mov_b64 s[20:21], 0x55\n s_mov_b64 vcc, 0x33" ::: "s20", "s21", "vcc");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 1) { REP64(asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cmp_lt_f32 s[20:21], %1, %4\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cmp_lt_f32 s[20:21], %2, %4\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cmp_lt_f32 s[20:21], %3, %4\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 2) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 3) { REP64(asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 4) { REP64(asm volatile("v_cndmask_b32_e32 %0, %4, %4, vcc\n v_cndmask_b32_e32 %1, %4, %4, vcc\n v_cndmask_b32_e32 %2, %4, %4, vcc\n v_cndmask_b32_e32 %3, %4, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 5) { REP64(asm volatile("v_cndmask_b32_e64 %0, %4, %4, s[20:21]\n v_cndmask_b32_e64 %1, %4, %4, s[20:21]\n v_cndmask_b32_e64 %2, %4, %4, s[20:21]\n v_cndmask_b32_e64 %3, %4, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 6) { REP64(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 7) { REP64(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 8) { REP64(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 9) { REP64(asm volatile("v_mul_f32_e64 %0, |%0|, %4\n v_mul_f32_e64 %1, |%1|, %4\n v_mul_f32_e64 %2, |%2|, %4\n v_mul_f32_e64 %3, |%3|, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 10) { REP64(asm volatile("v_mul_f32_e64 %0, %0, %4 clamp\n v_mul_f32_e64 %1, %1, %4 clamp\n v_mul_f32_e64 %2, %2, %4 clamp\n v_mul_f32_e64 %3, %3, %4 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 11) { REP64(asm volatile("v_mul_f32 %0, %4, %4\n v_mul_f32 %1, %4, %4\n v_mul_f32 %2, %4, %4\n v_mul_f32 %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 12) { REP64(asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 13) { REP64(asm volatile("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 14) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 15) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 16) { REP64(asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 17) { REP64(asm volatile("v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 7\n v_alignbit_b32 %3, %3, %3, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 18) { REP64(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 19) { REP64(asm volatile("v_sub_f32 %0, %5, %0\n v_sub_f32 %1, %5, %1\n v_sub_f32 %2, %5, %2\n v_sub_f32 %3, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 20) { REP64(asm volatile("v_fma_f32 %0, %0, %5, %4\n v_fma_f32 %1, %1, %5, %4\n v_fma_f32 %2, %2, %5, %4\n v_fma_f32 %3, %3, %5, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
        if (KIND == 21) { REP64(asm volatile("v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n v_mov_b32 %2, %2\n v_mov_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22");) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int KIND>
void run(const char* name, int per, float* out, float* in) {
    const int iters = 100;
    printf("%-40s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, in, 2, 1.0f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, in, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double insts_per_simd = (double)iters * 64 * 4 * wps * per;
        printf("  %dw %6.2f", wps, ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    printf("\n");
}
int main() {
    float *out, *in;
    (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    (void)hipMalloc(&in, 512 * 4);
    (void)hipMemset(in, 0, 512 * 4);
    printf("cycles (at 2.4 GHz) per wave64 instruction per SIMD, with 1/2/4/8 waves resident per SIMD\n");
    run<0>("cmp vcc ; cndmask vcc (pair)", 2, out, in);
    run<1>("cmp s[20:21] ; cndmask s[20:21] (pair)", 2, out, in);
    run<2>("cndmask_e64 vcc", 1, out, in);
    run<3>("cndmask_e32 vcc", 1, out, in);
    run<4>("cndmask_e32 vcc, distinct dst", 1, out, in);
    run<5>("cndmask_e64 s[20:21] distinct dst", 1, out, in);
    run<6>("v_xor_b32", 1, out, in);
    run<7>("v_or_b32", 1, out, in);
    run<8>("v_sub_u32", 1, out, in);
    run<9>("v_mul_f32 |v|", 1, out, in);
    run<10>("v_mul_f32 clamp", 1, out, in);
    run<11>("v_mul_f32 v,v,v distinct dst", 1, out, in);
    run<12>("v_add_f32 src0=src1", 1, out, in);
    run<13>("v_mul_legacy?/v_mul_i32_i24", 1, out, in);
    run<14>("v_mad_u32_u24", 1, out, in);
    run<15>("v_lshl_add_u32", 1, out, in);
    run<16>("v_add3_u32", 1, out, in);
    run<17>("v_alignbit_b32", 1, out, in);
    run<18>("v_cvt_f32_u32", 1, out, in);
    run<19>("v_sub_f32 v, s, v", 1, out, in);
    run<20>("v_fma_f32 v,v,s,v", 1, out, in);
    run<21>("v_accvgpr_write/read n/a", 1, out, in);
    return 0;
}

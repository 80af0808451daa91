// VALU issue-rate microbenchmark, part 3: select patterns after scalar writes of VCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, const float* in, int iters, float s0) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 64], a2 = a0 + 1, a3 = a1 + 1, x = a0 + 2;
    asm volatile("s_mov_b64 s[20:21], -1\n s_mov_b64 vcc, 0x33" ::: "s20", "s21", "vcc");
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n s_and_b64 vcc, vcc, s[20:21]\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22", "s23");) }
        if (KIND == 1) { REP64(asm volatile("v_cmp_lt_f32 s[22:23], %0, %4\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cndmask_b32_e64 %0, %0, %4, s[22:23]\n v_cndmask_b32_e64 %0, %0, %4, s[22:23]\n v_cndmask_b32_e64 %0, %0, %4, s[22:23]\n v_cmp_lt_f32 s[22:23], %1, %4\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]\n v_cmp_lt_f32 s[22:23], %2, %4\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cmp_lt_f32 s[22:23], %3, %4\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]\n v_cndmask_b32_e64 %3, %3, %4, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22", "s23");) }
        if (KIND == 2) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_and_saveexec_b64 s[22:23], vcc\n v_mov_b32 %0, %4\n v_mov_b32 %0, %4\n v_mov_b32 %0, %4\n s_or_b64 exec, exec, s[22:23]\n v_cmp_lt_f32 vcc, %1, %4\n s_and_saveexec_b64 s[22:23], vcc\n v_mov_b32 %1, %4\n v_mov_b32 %1, %4\n v_mov_b32 %1, %4\n s_or_b64 exec, exec, s[22:23]\n v_cmp_lt_f32 vcc, %2, %4\n s_and_saveexec_b64 s[22:23], vcc\n v_mov_b32 %2, %4\n v_mov_b32 %2, %4\n v_mov_b32 %2, %4\n s_or_b64 exec, exec, s[22:23]\n v_cmp_lt_f32 vcc, %3, %4\n s_and_saveexec_b64 s[22:23], vcc\n v_mov_b32 %3, %4\n v_mov_b32 %3, %4\n v_mov_b32 %3, %4\n s_or_b64 exec, exec, s[22:23]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22", "s23");) }
        if (KIND == 3) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22", "s23");) }
        if (KIND == 4) { REP64(asm volatile("s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %0, %0, %4\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_mul_f32 %1, %1, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %1, %1, %4\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_mul_f32 %2, %2, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %2, %2, %4\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_mul_f32 %3, %3, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(s0) : "vcc", "s20", "s21", "s22", "s23");) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int KIND>
void run(const char* name, int per, float* out, float* in) {
    const int iters = 100;
    printf("%-58s", name);
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
    printf("cycles (at 2.4 GHz) per wave64 VALU instruction per SIMD, with 1/2/4/8 waves resident per SIMD\n");
    run<0>("cmp vcc; s_and vcc; 3x cndmask_e32 vcc (per 4 valu)", 4, out, in);
    run<1>("cmp s[22:23]; s_and s[22:23]; 3x cndmask_e64 (per 4 valu)", 4, out, in);
    run<2>("cmp vcc; s_and_saveexec; 3x v_mov; s_or exec (per 4 valu)", 4, out, in);
    run<3>("cmp vcc; 3x cndmask_e32 vcc (per 4 valu)", 4, out, in);
    run<4>("s_mov vcc; 1x cndmask_e32 vcc; 3x v_mul (per 4 valu)", 4, out, in);
    return 0;
}

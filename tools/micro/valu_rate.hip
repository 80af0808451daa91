// Issue-rate microbenchmark for the vector ALU of gfx950: cycles per wave64 instruction per SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, float s0, float s1) {
    float a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], a2 = a0 + 1, a3 = a1 + 1, a4 = a0 + 2, a5 = a1 + 2, a6 = a0 + 3, a7 = a1 + 3;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    v2f q0 = {a1, a0}, q1 = {a3, a2};
    v2f ss = {s0, s1};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // v_mul_f32 vgpr,vgpr: 4 independent chains
            REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (KIND == 1) {  // v_pk_mul_f32 vgpr pairs only
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0));)
        } else if (KIND == 2) {  // v_pk_mul_f32 with an SGPR pair operand
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(ss));)
        } else if (KIND == 3) {  // v_pk_add_f32 vgpr
            REP64(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0));)
        } else if (KIND == 4) {  // v_pk_mul_f32 d = a*b with three different register pairs (no operand reuse)
            REP64(asm volatile("v_pk_mul_f32 %0, %1, %4\n v_pk_mul_f32 %1, %2, %5\n v_pk_mul_f32 %2, %3, %4\n v_pk_mul_f32 %3, %0, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1));)
        } else if (KIND == 5) {  // v_pk_fma_f32
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0));)
        } else if (KIND == 6) {  // v_mul_f32 with an SGPR operand
            REP64(asm volatile("v_mul_f32 %0, %4, %0\n v_mul_f32 %1, %4, %1\n v_mul_f32 %2, %4, %2\n v_mul_f32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0));)
        } else if (KIND == 7) {  // v_cmp_lt_f32 into vcc
            REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (KIND == 8) {  // v_cmp_lt_f32 into sgpr pair (VOP3)
            REP64(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[20:21], %2, %3\n v_cmp_lt_f32 s[22:23], %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");)
        } else if (KIND == 9) {  // v_cndmask
            REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (KIND == 10) {  // v_mov
            REP64(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 11) {  // v_rcp_f32
            REP64(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 12) {  // v_fma_f32
            REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));)
        } else if (KIND == 13) {  // v_pk_mul_f32 with op_sel broadcast of a vgpr
            REP64(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char* name, float* out, float* in) {
    const int iters = 200, blocks = 256 * 8;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, 2, 1.0f, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, in, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double insts_per_simd = (double)iters * 64 * 4 * 8;  // per wave x 8 waves on the SIMD
    double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-44s %7.3f ms  %.2f cycles per wave-instruction per SIMD (clock %d MHz)\n", name, ms, cycles / insts_per_simd, clk_khz / 1000);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&in, 512 * 4);
    hipMemset(in, 0, 512 * 4);
    run<0>("v_mul_f32 v,v,v", out, in);
    run<6>("v_mul_f32 v,s,v", out, in);
    run<12>("v_fma_f32 v,v,v,v", out, in);
    run<1>("v_pk_mul_f32 v,v,v (dst=src0)", out, in);
    run<4>("v_pk_mul_f32 v,v,v (distinct)", out, in);
    run<2>("v_pk_mul_f32 v,v,s", out, in);
    run<13>("v_pk_mul_f32 v,v,v op_sel_hi broadcast", out, in);
    run<3>("v_pk_add_f32 v,v,v", out, in);
    run<5>("v_pk_fma_f32 v,v,v,v", out, in);
    run<7>("v_cmp_lt_f32 vcc", out, in);
    run<8>("v_cmp_lt_f32 sgpr", out, in);
    run<9>("v_cndmask_b32", out, in);
    run<10>("v_mov_b32", out, in);
    run<11>("v_rcp_f32", out, in);
    return 0;
}

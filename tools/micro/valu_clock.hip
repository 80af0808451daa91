// Vector-ALU issue cost on gfx950 measured INSIDE the kernel: every wave brackets a run of independent instructions of one kind
// with s_memtime (the shader clock counter) and with s_memrealtime (a constant 100 MHz counter), so the result does not depend on
// the nominal clock or on launch overhead: cycles per wave64 instruction per SIMD = shader-clock ticks x resident waves per SIMD /
// instructions per wave, and the two counters together give the clock the chip actually ran at.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_clock.hip -o tools/micro/valu_clock && tools/micro/valu_clock
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

enum { MUL_VV, FMA_VVV, MUL_VS, MAX_VV, MIN3, CMP, CNDMASK, CNDMASK_OTHER_DST, CNDMASK_SGPR, CNDMASK_CONST, CNDMASK_E64_VCC, CNDMASK_VALU_VCC, CMP_THEN_CNDMASK, CMP_3CND, CMP_3CND_E64, CND_MUL, SALU_VCC_CND_3MUL, PK_MUL, PK_FMA, PK_FMA_S, RCP, SQRT, MOV, ADD_U32, KINDS };
static const char* kNames[KINDS] = {"v_mul_f32 v,v,v", "v_fma_f32 v,v,v,v", "v_mul_f32 v,s,v", "v_max_f32 v,v,v", "v_min3_f32 v,v,v,v", "v_cmp_lt_f32 vcc,v,v",
                                    "v_cndmask_b32 v,v,v,vcc", "v_cndmask_b32 w,v,v,vcc", "v_cndmask_b32 v,v,v,s[a:b]", "v_cndmask_b32 v,0,1.0,vcc", "v_cndmask_b32_e64 v,v,v,vcc", "v_cndmask e32, vcc from v_cmp", "v_cmp + v_cndmask e32 (pair)", "v_cmp + 3 v_cndmask e32", "v_cmp + 3 v_cndmask e64", "v_cndmask e32, v_mul (pair)", "s_mov vcc, cndmask e32, 2 mul", "v_pk_mul_f32 v,v,v", "v_pk_fma_f32 v,v,v,v", "v_pk_fma_f32 v,s,v,v", "v_rcp_f32", "v_sqrt_f32",
                                    "v_mov_b32 v,v", "v_add_u32 v,v,v"};

// 4 independent accumulators, 32 x 4 = 128 instructions per loop turn
#define BODY(text) REP32(asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(x), "s"(s0), "v"(px), "s"(ss) : "vcc", "s20", "s21");)

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* ticks, const float* in, int iters, float s0) {
    float a0 = in[threadIdx.x], a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, x = a0 + 1.5f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0}, px = {x, x};
    v2f ss = {s0, s0};
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memrealtime %0\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0)::"memory");
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n s_mov_b32 s20, 0x33333333\n s_mov_b32 s21, 0x33333333" ::: "vcc", "s20", "s21");  // defined masks for the selects
    if (KIND == CNDMASK_VALU_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a0), "v"(x) : "vcc");  // the mask written by the vector ALU
    for (int i = 0; i < iters; ++i) {
        if (KIND == MUL_VV) { BODY("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8") }
        if (KIND == FMA_VVV) { BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8") }
        if (KIND == MUL_VS) { BODY("v_mul_f32 %0, %9, %0\n v_mul_f32 %1, %9, %1\n v_mul_f32 %2, %9, %2\n v_mul_f32 %3, %9, %3") }
        if (KIND == MAX_VV) { BODY("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8") }
        if (KIND == MIN3) { BODY("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8") }
        if (KIND == CMP) { BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8") }
        if (KIND == CNDMASK) { BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc") }
        if (KIND == CNDMASK_OTHER_DST) { BODY("v_cndmask_b32 %0, %1, %8, vcc\n v_cndmask_b32 %1, %2, %8, vcc\n v_cndmask_b32 %2, %3, %8, vcc\n v_cndmask_b32 %3, %0, %8, vcc") }
        if (KIND == CNDMASK_SGPR) { BODY("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]") }
        if (KIND == CNDMASK_CONST) { BODY("v_cndmask_b32_e64 %0, 0, 1.0, vcc\n v_cndmask_b32_e64 %1, 0, 1.0, vcc\n v_cndmask_b32_e64 %2, 0, 1.0, vcc\n v_cndmask_b32_e64 %3, 0, 1.0, vcc") }
        if (KIND == CNDMASK_E64_VCC) { BODY("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc") }
        if (KIND == CNDMASK_VALU_VCC) { BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc") }
        if (KIND == CMP_THEN_CNDMASK) { BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc") }
        if (KIND == CMP_3CND) { BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc") }
        if (KIND == CMP_3CND_E64) { BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc") }
        if (KIND == CND_MUL) { BODY("v_cndmask_b32 %0, %0, %8, vcc\n v_mul_f32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_mul_f32 %3, %3, %8") }
        if (KIND == SALU_VCC_CND_3MUL) { BODY("s_mov_b64 vcc, exec\n v_cndmask_b32 %0, %0, %8, vcc\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8") }
        if (KIND == PK_MUL) { BODY("v_pk_mul_f32 %4, %4, %10\n v_pk_mul_f32 %5, %5, %10\n v_pk_mul_f32 %6, %6, %10\n v_pk_mul_f32 %7, %7, %10") }
        if (KIND == PK_FMA) { BODY("v_pk_fma_f32 %4, %4, %10, %10\n v_pk_fma_f32 %5, %5, %10, %10\n v_pk_fma_f32 %6, %6, %10, %10\n v_pk_fma_f32 %7, %7, %10, %10") }
        if (KIND == PK_FMA_S) { BODY("v_pk_fma_f32 %4, %11, %4, %10\n v_pk_fma_f32 %5, %11, %5, %10\n v_pk_fma_f32 %6, %11, %6, %10\n v_pk_fma_f32 %7, %11, %7, %10") }
        if (KIND == RCP) { BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3") }
        if (KIND == SQRT) { BODY("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3") }
        if (KIND == MOV) { BODY("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8") }
        if (KIND == ADD_U32) { BODY("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8") }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (threadIdx.x == 0) {  // (a vector store: one record per wave)
        ticks[4 * blockIdx.x] = t1 - t0;
        ticks[4 * blockIdx.x + 1] = r1 - r0;
        ticks[4 * blockIdx.x + 2] = r0;  // absolute 100 MHz times: when the whole grid began and ended
        ticks[4 * blockIdx.x + 3] = r1;
    }
    if (a0 + a1 + a2 + a3 + p0.x + p1.x + p2.x + p3.x == 12345.678f) ticks[0] = 0;  // keeps the accumulators alive
}

template <int KIND>
static void run(unsigned long long* d_ticks, float* d_in, int cus) {
    const int iters = 2000;                         // 2000 x 128 = 256 000 instructions per wave
    const double instr = (double)iters * 128.0;
    std::printf("%-26s", kNames[KIND]);
    for (int wps = 1; wps <= 8; wps *= 2) {         // resident waves per SIMD (the grid is one round of the chip: cus x 4 x wps single-wave blocks)
        const int blocks = cus * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_ticks, d_in, 8, 1.0f);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_ticks, d_in, iters, 1.0f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(4 * blocks);
        (void)hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc(blocks), mhz(blocks);
        unsigned long long first = ~0ull, last = 0;
        for (int b = 0; b < blocks; ++b) {
            cyc[b] = (double)h[4 * b];
            mhz[b] = h[4 * b + 1] ? 100.0 * (double)h[4 * b] / (double)h[4 * b + 1] : 0.0;
            first = std::min(first, h[4 * b + 2]);
            last = std::max(last, h[4 * b + 3]);
        }
        std::nth_element(cyc.begin(), cyc.begin() + blocks / 2, cyc.end());
        std::nth_element(mhz.begin(), mhz.begin() + blocks / 2, mhz.end());
        // a SIMD issues for its `wps` waves in turn: the cost of one instruction is the wave's elapsed ticks / (instr x wps)
        // and for the chip as a whole: the span from the first wave's start to the last wave's end, in shader-clock cycles, times the
        // SIMDs, over all instructions issued - equal to the per-wave figure when all waves really ran side by side
        const double span_cycles = (double)(last - first) / 100.0 * mhz[blocks / 2];
        std::printf("  %dw %5.2f|%5.2f (%4.0f MHz)", wps, cyc[blocks / 2] / (instr * wps), span_cycles * (cus * 4.0) / (instr * blocks), mhz[blocks / 2]);
    }
    std::printf("\n");
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* d_ticks;
    float* d_in;
    (void)hipMalloc(&d_ticks, (size_t)cus * 4 * 8 * 4 * 8);
    (void)hipMalloc(&d_in, 64 * 4);
    (void)hipMemset(d_in, 0, 64 * 4);
    std::printf("%d compute units; median over waves of: s_memtime ticks per wave64 instruction per SIMD with 1/2/4/8 waves per SIMD\n"
                "after the bar: the same from the whole grid's span (first start to last end); in brackets: s_memtime ticks per microsecond of\n"
                "s_memrealtime = the rate of that counter\n", cus);
    run<MUL_VV>(d_ticks, d_in, cus);
    run<FMA_VVV>(d_ticks, d_in, cus);
    run<MUL_VS>(d_ticks, d_in, cus);
    run<MAX_VV>(d_ticks, d_in, cus);
    run<MIN3>(d_ticks, d_in, cus);
    run<CMP>(d_ticks, d_in, cus);
    run<CNDMASK>(d_ticks, d_in, cus);
    run<CNDMASK_OTHER_DST>(d_ticks, d_in, cus);
    run<CNDMASK_SGPR>(d_ticks, d_in, cus);
    run<CNDMASK_CONST>(d_ticks, d_in, cus);
    run<CNDMASK_E64_VCC>(d_ticks, d_in, cus);
    run<CNDMASK_VALU_VCC>(d_ticks, d_in, cus);
    run<CMP_THEN_CNDMASK>(d_ticks, d_in, cus);
    run<CMP_3CND>(d_ticks, d_in, cus);
    run<CMP_3CND_E64>(d_ticks, d_in, cus);
    run<CND_MUL>(d_ticks, d_in, cus);
    run<SALU_VCC_CND_3MUL>(d_ticks, d_in, cus);
    run<PK_MUL>(d_ticks, d_in, cus);
    run<PK_FMA>(d_ticks, d_in, cus);
    run<PK_FMA_S>(d_ticks, d_in, cus);
    run<RCP>(d_ticks, d_in, cus);
    run<SQRT>(d_ticks, d_in, cus);
    run<MOV>(d_ticks, d_in, cus);
    run<ADD_U32>(d_ticks, d_in, cus);
    return 0;
}

// Does a vector instruction on gfx950 cost less when lanes are masked off?  Every wave runs a run of independent instructions of one
// kind under a given EXEC mask and brackets it with s_memtime (shader clock), as tools/micro/valu_clock.hip does; eight resident waves
// per SIMD, one round of the chip.  Prints cycles per wave64 instruction per SIMD for each mask.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_exec.hip -o tools/micro/valu_exec && tools/micro/valu_exec
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

enum { MUL_VV, FMA_VVV, MAX_VV, CMP, PK_FMA, PK_FMA_S, RCP, KINDS };
static const char* kNames[KINDS] = {"v_mul_f32 v,v,v", "v_fma_f32 v,v,v,v", "v_max_f32 v,v,v", "v_cmp_lt_f32 vcc,v,v", "v_pk_fma_f32 v,v,v,v", "v_pk_fma_f32 v,s,v,v", "v_rcp_f32"};

#define BODY(text) REP32(asm volatile(text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(x), "s"(s0), "v"(px), "s"(ss) : "vcc");)

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* ticks, const float* in, int iters, float s0, unsigned long long mask) {
    float a0 = in[threadIdx.x], a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, x = a0 + 1.5f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0}, px = {x, x};
    v2f ss = {s0, s0};
    unsigned long long t0, t1, saved;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_mov_b64 %0, exec\n s_mov_b64 exec, %2\n s_memtime %1\n s_waitcnt lgkmcnt(0)" : "=&s"(saved), "=s"(t0) : "s"(mask) : "memory");
    for (int i = 0; i < iters; ++i) {
        if (KIND == MUL_VV) { BODY("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8") }
        if (KIND == FMA_VVV) { BODY("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8") }
        if (KIND == MAX_VV) { BODY("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8") }
        if (KIND == CMP) { BODY("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8") }
        if (KIND == PK_FMA) { BODY("v_pk_fma_f32 %4, %4, %10, %10\n v_pk_fma_f32 %5, %5, %10, %10\n v_pk_fma_f32 %6, %6, %10, %10\n v_pk_fma_f32 %7, %7, %10, %10") }
        if (KIND == PK_FMA_S) { BODY("v_pk_fma_f32 %4, %11, %4, %10\n v_pk_fma_f32 %5, %11, %5, %10\n v_pk_fma_f32 %6, %11, %6, %10\n v_pk_fma_f32 %7, %11, %7, %10") }
        if (KIND == RCP) { BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3") }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, %1" : "=s"(t1) : "s"(saved) : "memory");
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;  // (a vector store: one record per wave)
    if (a0 + a1 + a2 + a3 + p0.x + p1.x + p2.x + p3.x == 12345.678f) ticks[0] = 0;  // keeps the accumulators alive
}

static const unsigned long long kMasks[] = {~0ull, 0x00000000ffffffffull, 0xffffffff00000000ull, 0x000000000000ffffull, 0x0000ffff0000ffffull,
                                            0x5555555555555555ull, 0x0000000000000001ull, 0x0001000100010001ull, 0x00000000ffff0001ull};
static const char* kMaskNames[] = {"all 64", "low 32", "high 32", "low 16", "quarters 0+2", "even lanes", "lane 0", "1 per quarter", "lane 0 + q1"};

template <int KIND>
static void run(unsigned long long* d_ticks, float* d_in, int cus) {
    const int iters = 1000, wps = 8;
    const double instr = (double)iters * 128.0;
    const int blocks = cus * 4 * wps;
    std::printf("%-24s", kNames[KIND]);
    for (unsigned m = 0; m < sizeof(kMasks) / sizeof(kMasks[0]); ++m) {
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_ticks, d_in, 8, 1.0f, kMasks[m]);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_ticks, d_in, iters, 1.0f, kMasks[m]);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), d_ticks, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        std::printf(" %6.2f", (double)h[blocks / 2] * wps / instr);
    }
    std::printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_ticks = nullptr;
    float* d_in = nullptr;
    if (hipMalloc((void**)&d_ticks, (size_t)cus * 4 * 8 * 8) != hipSuccess || hipMalloc((void**)&d_in, 64 * 4) != hipSuccess) return 1;
    std::vector<float> in(64, 1.0f);
    (void)hipMemcpy(d_in, in.data(), 256, hipMemcpyHostToDevice);
    std::printf("cycles per wave64 instruction per SIMD (8 resident waves per SIMD, median wave) under EXEC masks:\n%-24s", "");
    for (unsigned m = 0; m < sizeof(kMasks) / sizeof(kMasks[0]); ++m) std::printf(" %s |", kMaskNames[m]);
    std::printf("\n");
    run<MUL_VV>(d_ticks, d_in, cus);
    run<FMA_VVV>(d_ticks, d_in, cus);
    run<MAX_VV>(d_ticks, d_in, cus);
    run<CMP>(d_ticks, d_in, cus);
    run<PK_FMA>(d_ticks, d_in, cus);
    run<PK_FMA_S>(d_ticks, d_in, cus);
    run<RCP>(d_ticks, d_in, cus);
    (void)hipFree(d_ticks);
    (void)hipFree(d_in);
    return 0;
}

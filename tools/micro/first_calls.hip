// What a process's first HIP calls cost: context, first allocation, first copies (pageable, pinned, by a kernel), first kernel
// launch, a code object with 128 MiB of zero-initialised data.  The argument picks which kind of copy comes first:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/first_calls.hip -o tools/micro/first_calls && tools/micro/first_calls [pageable|pinned|kernel|async]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_copy(const unsigned* from, unsigned* to, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) to[i] = from[i];
}

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STEP(name, expr)                                  \
    do {                                                  \
        const double t0 = now();                          \
        hipError_t e = (expr);                            \
        printf("%-46s %9.3f ms  %s\n", name, now() - t0, e == hipSuccess ? "" : hipGetErrorString(e)); \
    } while (0)

int main(int argc, char** argv) {
    const char* first = argc > 1 ? argv[1] : "pageable";
    void *a = nullptr, *c = nullptr, *pin = nullptr;
    std::vector<char> host(1 << 20, 1);
    STEP("hipFree(0) (context)", hipFree(nullptr));
    STEP("first hipMalloc(16)", hipMalloc(&a, 16));
    STEP("hipMalloc(1 MiB)", hipMalloc(&c, 1 << 20));
    STEP("hipHostMalloc 1 MiB", hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault));
    std::memset(pin, 1, 1 << 20);
    for (int round = 0; round < 2; ++round) {
        if ((round == 0) == (std::strcmp(first, "pinned") == 0)) STEP("hipMemcpy H2D 4096 B pinned", hipMemcpy(c, pin, 4096, hipMemcpyHostToDevice));
        if ((round == 0) == (std::strcmp(first, "async") == 0)) {
            const double t0 = now();
            hipMemcpyAsync(c, pin, 4096, hipMemcpyHostToDevice, nullptr);
            hipStreamSynchronize(nullptr);
            printf("%-46s %9.3f ms\n", "hipMemcpyAsync H2D 4096 B pinned + sync", now() - t0);
        }
        if ((round == 0) == (std::strcmp(first, "kernel") == 0)) {
            const double t0 = now();
            hipLaunchKernelGGL(k_copy, dim3(4), dim3(256), 0, nullptr, (const unsigned*)pin, (unsigned*)c, 1024u);
            hipError_t e = hipDeviceSynchronize();
            printf("%-46s %9.3f ms  %s\n", "kernel reads 4096 B of pinned host memory", now() - t0, e == hipSuccess ? "" : hipGetErrorString(e));
        }
        if ((round == 0) == (std::strcmp(first, "pageable") == 0)) STEP("hipMemcpy H2D 16 B pageable", hipMemcpy(a, host.data(), 16, hipMemcpyHostToDevice));
    }
    STEP("hipMemcpy H2D 1 MiB pageable", hipMemcpy(c, host.data(), 1 << 20, hipMemcpyHostToDevice));
    STEP("hipMemcpy H2D 1 MiB pinned", hipMemcpy(c, pin, 1 << 20, hipMemcpyHostToDevice));
    STEP("hipMemcpy D2H 1 MiB pinned", hipMemcpy(pin, c, 1 << 20, hipMemcpyDeviceToHost));
    STEP("hipMemcpy D2H 4096 B pageable", hipMemcpy(host.data(), c, 4096, hipMemcpyDeviceToHost));
    return 0;
}

// Where a fresh process's first HIP milliseconds go (compile on the GPU box: hipcc --offload-arch=gfx950 -O2 profiles/startup_probe.hip -o /tmp/startup_probe).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(int* p) { if (threadIdx.x == 0) p[0] = 1; }
#define STEP(what, ...) do { const double t0 = now(); hipError_t e = (__VA_ARGS__); std::printf("%-44s %8.2f ms%s\n", what, now() - t0, e == hipSuccess ? "" : "  FAILED"); } while (0)
int main()
{
    const double t00 = now();
    STEP("hipSetDevice(0)", hipSetDevice(0));
    STEP("hipFree(0) (context)", hipFree(nullptr));
    hipStream_t s1, s2, s3; int* d = nullptr; void* h = nullptr; void* h2 = nullptr;
    STEP("hipStreamCreateWithFlags #1", hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    STEP("hipStreamCreateWithFlags #2", hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    STEP("hipStreamCreateWithFlags #3", hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    STEP("hipMalloc 64 MB", hipMalloc((void**)&d, 64 << 20));
    STEP("hipHostMalloc 8 MB", hipHostMalloc(&h, 8 << 20));
    STEP("hipHostMalloc 32 MB", hipHostMalloc(&h2, 32 << 20));
    STEP("hipMemsetAsync + sync on #1", [&] { (void)hipMemsetAsync(d, 0, 64 << 20, s1); return hipStreamSynchronize(s1); }());
    STEP("first kernel launch + sync on #1", [&] { touch<<<dim3(1), dim3(64), 0, s1>>>(d); return hipStreamSynchronize(s1); }());
    STEP("second kernel launch + sync on #2", [&] { touch<<<dim3(1), dim3(64), 0, s2>>>(d); return hipStreamSynchronize(s2); }());
    STEP("memset on the null stream + sync", [&] { (void)hipMemset(d, 0, 1024); return hipDeviceSynchronize(); }());
    std::printf("%-44s %8.2f ms\n", "total", now() - t00);
    return 0;
}

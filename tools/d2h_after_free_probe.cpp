// developer tool: which release puts the process into the 0.4 s of slow PCIe copies that bench.settle_copies waits out.
//   hipcc -O2 --offload-arch=gfx950 tools/d2h_after_free_probe.cpp -o build/d2h_after_free && ./build/d2h_after_free
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static void *dev, *pin;
static const size_t NB = 4800000;
static double rate()
{
    (void)hipMemcpy(pin, dev, NB, hipMemcpyDeviceToHost);
    auto t0 = std::chrono::steady_clock::now();
    (void)hipMemcpy(pin, dev, NB, hipMemcpyDeviceToHost);
    return NB / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
}
static void watch(const char* what)
{
    printf("%-46s", what);
    auto t0 = std::chrono::steady_clock::now();
    double first = rate(), slow_until = -1;
    for (int k = 0; k < 40; ++k) {
        const double r = rate();
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (r < 35.0) slow_until = t;
        std::this_thread::sleep_for(std::chrono::milliseconds(25));
    }
    printf(" first copy %.1f GB/s, slow until %.2f s\n", first, slow_until);
}
int main()
{
    CK(hipMalloc(&dev, NB)); CK(hipHostMalloc(&pin, NB, hipHostMallocDefault));
    watch("quiet process");
    void* p;
    CK(hipMalloc(&p, 4096)); CK(hipFree(p)); watch("hipFree of 4 KB");
    CK(hipMalloc(&p, 64u << 20)); CK(hipFree(p)); watch("hipFree of 64 MB");
    CK(hipMalloc(&p, (size_t)4 << 30)); CK(hipMemset(p, 0, (size_t)4 << 30)); CK(hipDeviceSynchronize()); CK(hipFree(p)); watch("hipFree of 4 GB (written)");
    CK(hipHostMalloc(&p, 4096, 0)); CK(hipHostFree(p)); watch("hipHostFree of 4 KB");
    CK(hipHostMalloc(&p, 256u << 20, 0)); CK(hipHostFree(p)); watch("hipHostFree of 256 MB");
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipMemsetAsync(dev, 0, 64, s)); CK(hipStreamSynchronize(s)); CK(hipStreamDestroy(s)); watch("hipStreamDestroy of a used stream");
    hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, 0)); CK(hipEventSynchronize(ev)); CK(hipEventDestroy(ev)); watch("hipEventDestroy");
    CK(hipMalloc(&p, (size_t)1 << 30)); watch("hipMalloc of 1 GB (kept)");
    return 0;
}

// developer tool: device-to-host copy rate of one 58 MB table per STREAM, streams created one after the other (the rate a
// communicator's stream gets depends on which copy engine the runtime gives it).   hipcc -O2 tools/d2h_stream_probe.cpp -o build/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const size_t bytes = 58u << 20;
    const int ns = argc > 1 ? atoi(argv[1]) : 10;
    void *dev, *pin;
    CK(hipMalloc(&dev, bytes)); CK(hipMemset(dev, 1, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    std::vector<hipStream_t> st(ns);
    for (int i = 0; i < ns; ++i) {
        CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        double best = 1e9, bestp = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st[i]));
            CK(hipStreamSynchronize(st[i]));
            double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep && t < best) best = t;
            // the same bytes as 23 pieces
            t0 = std::chrono::steady_clock::now();
            const size_t piece = bytes / 23 / 256 * 256;
            for (int k = 0; k < 23; ++k) CK(hipMemcpyAsync((char*)pin + k * piece, (char*)dev + k * piece, piece, hipMemcpyDeviceToHost, st[i]));
            CK(hipStreamSynchronize(st[i]));
            t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep && t < bestp) bestp = t;
        }
        printf("stream %d: one copy %.2f ms (%.1f GB/s), 23 pieces %.2f ms (%.1f GB/s)\n", i, best * 1e3, bytes / best / 1e9, bestp * 1e3, bytes / bestp / 1e9);
    }
    // the null stream and a fresh page-locked buffer
    void* pin2; CK(hipHostMalloc(&pin2, bytes, hipHostMallocDefault));
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMemcpy(pin2, dev, bytes, hipMemcpyDeviceToHost));
        double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("null stream, second buffer: %.2f ms (%.1f GB/s)\n", t * 1e3, bytes / t / 1e9);
    }
    return 0;
}

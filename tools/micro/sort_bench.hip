// micro-benchmark: rocPRIM onesweep radix sort of (u64 key, u32 value) pairs with different digit widths
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
typedef unsigned long long u64; typedef unsigned int u32;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <class Config> float run(const char* name, u64* kin, u64* kout, u32* vin, u32* vout, size_t n, int b0, int b1)
{
    size_t bytes = 0; void* tmp = nullptr;
    CK(rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, b0, b1, 0));
    CK(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) CK(rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, b0, b1, 0));
    CK(hipEventRecord(e0, 0));
    const int R = 10;
    for (int r = 0; r < R; ++r) CK(rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, b0, b1, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s bits[%d,%d): %.1f us/sort (tmp %zu KB)\n", name, b0, b1, ms / R * 1e3, bytes >> 10);
    CK(hipFree(tmp));
    return ms / R;
}
int main(int argc, char** argv)
{
    size_t n = argc > 1 ? atol(argv[1]) : 5000000;
    std::vector<u64> h(n); std::vector<u32> v(n);
    u64 x = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x & ((1ull << 56) - 1); v[i] = (u32)i; }
    u64 *kin, *kout; u32 *vin, *vout;
    CK(hipMalloc(&kin, n * 8)); CK(hipMalloc(&kout, n * 8)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    CK(hipMemcpy(kin, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(vin, v.data(), n * 4, hipMemcpyHostToDevice));
    using namespace rocprim;
    run<default_config>("default", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<512, 12>, kernel_config<512, 12>, 9, block_radix_rank_algorithm::match>>>("os 512x12 r9", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<512, 12>, kernel_config<512, 12>, 10, block_radix_rank_algorithm::match>>>("os 512x12 r10", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 8>, 9, block_radix_rank_algorithm::match>>>("os 1024x8 r9", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 8>, kernel_config<1024, 8>, 10, block_radix_rank_algorithm::match>>>("os 1024x8 r10", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<512, 16>, kernel_config<512, 16>, 9, block_radix_rank_algorithm::match>>>("os 512x16 r9", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<512, 10>, kernel_config<512, 10>, 11, block_radix_rank_algorithm::match>>>("os 512x10 r11", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024, 6>, kernel_config<1024, 6>, 10, block_radix_rank_algorithm::match>>>("os 1024x6 r10", kin, kout, vin, vout, n, 11, 56);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 8>, kernel_config<256, 8>, 10, block_radix_rank_algorithm::match>>>("os 256x8 r10", kin, kout, vin, vout, n, 11, 56);
    return 0;
}

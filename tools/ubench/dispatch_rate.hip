// How fast does the chip START the waves of one launch?  An (almost) empty kernel with the headline kernel's launch
// shape -- 1152 workgroups of 256 threads, 31 KB of dynamic LDS, 120 VGPRs -- and variations of it; every wave records
// the chip-wide 100 MHz counter at its first instruction and spins for a fixed time.  Prints, per variant, when the
// last wave started (the dispatch ramp) and the launch's duration between HIP events.  Not part of the product.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

extern __shared__ unsigned char lds[];

template <int VGPRS>
__global__ void __launch_bounds__(1024) k_spin(unsigned long long* starts, int spin_ticks, int touch_lds)
{
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    if (VGPRS > 64) {  // keep a high register alive so that the wave is allocated VGPRS registers
        float x = float(threadIdx.x);
        if (VGPRS >= 120) asm volatile("v_mov_b32 v119, %0" ::"v"(x) : "v119");
        else asm volatile("v_mov_b32 v95, %0" ::"v"(x) : "v95");
    }
    if (touch_lds) lds[threadIdx.x] = (unsigned char)threadIdx.x;
    unsigned long long t1 = t0;
    while (t1 - t0 < (unsigned long long)spin_ticks) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) starts[(size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6] = t0;
}

// the same with what the real kernels add at a wave's start: two big by-value argument structs (their scalar loads are
// awaited before the stamp, as in the kernels' stamps build) and a long straight-line body (instruction-cache fills)
struct BigArgs {
    unsigned long long* starts;
    int spin_ticks;
    int pad[110];
    int last;
};
template <int BODY>
__global__ void __launch_bounds__(256) k_spin_big(const BigArgs a, const BigArgs b)
{
    const int sum = a.last + b.last + a.pad[50] + b.pad[100];  // fields at the far end of both structs
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : "s"(sum) : "memory");
    float x = float(threadIdx.x) + float(sum);
    asm volatile("v_mov_b32 v119, %0" ::"v"(x) : "v119");
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    if (BODY) {
#pragma unroll
        for (int i = 0; i < BODY; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));  // 8 bytes each, executed once
    }
    unsigned long long t1 = t0;
    while (t1 - t0 < (unsigned long long)a.spin_ticks) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) a.starts[(size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6] = t0 + (x == 12345.f ? 1 : 0);
}

template <int BODY>
int run_big(const char* name, int wgs, int threads, int lds_bytes, int spin_ticks)
{
    const int waves = wgs * threads / 64;
    unsigned long long* d;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * waves));
    CHECK(hipFuncSetAttribute((const void*)k_spin_big<BODY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    BigArgs a{};
    a.starts = d;
    a.spin_ticks = spin_ticks;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_spin_big<BODY>, dim3(wgs), dim3(threads), lds_bytes, 0, a, a);
    CHECK(hipDeviceSynchronize());
    float best_ms = 1e9f;
    std::vector<unsigned long long> h(waves);
    double last_start = 0, p50 = 0, p90 = 0;
    for (int rep = 0; rep < 10; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin_big<BODY>, dim3(wgs), dim3(threads), lds_bytes, 0, a, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_ms) {
            best_ms = ms;
            CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            last_start = double(h.back() - h.front()) * 0.01;
            p50 = double(h[waves / 2] - h.front()) * 0.01;
            p90 = double(h[waves * 9 / 10] - h.front()) * 0.01;
        }
    }
    printf("%-58s %5d wgs x %4d thr  lds %6d  spin %4.1f us: launch %6.2f us, wave start median %5.2f / p90 %5.2f / last %5.2f us\n",
           name, wgs, threads, lds_bytes, spin_ticks * 0.01, best_ms * 1e3, p50, p90, last_start);
    CHECK(hipFree(d));
    return 0;
}

template <int VGPRS>
int run(const char* name, int wgs, int threads, int lds_bytes, int spin_ticks)
{
    const int waves = wgs * threads / 64;
    unsigned long long* d;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * waves));
    CHECK(hipFuncSetAttribute((const void*)k_spin<VGPRS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_spin<VGPRS>, dim3(wgs), dim3(threads), lds_bytes, 0, d, spin_ticks, 1);
    CHECK(hipDeviceSynchronize());
    float best_ms = 1e9f;
    std::vector<unsigned long long> h(waves);
    double last_start = 0, p50 = 0;
    for (int rep = 0; rep < 10; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spin<VGPRS>, dim3(wgs), dim3(threads), lds_bytes, 0, d, spin_ticks, 1);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_ms) {
            best_ms = ms;
            CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            last_start = double(h.back() - h.front()) * 0.01;
            p50 = double(h[waves / 2] - h.front()) * 0.01;
        }
    }
    printf("%-58s %5d wgs x %4d thr  lds %6d  spin %4.1f us: launch %6.2f us, median wave start %5.2f us, last wave start %5.2f us\n",
           name, wgs, threads, lds_bytes, spin_ticks * 0.01, best_ms * 1e3, p50, last_start);
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    // the headline launch shape and what changes the ramp
    run<120>("headline shape (4-wave wgs, 31 KB LDS, 120 VGPRs)", 1152, 256, 31232, 100);
    run<120>("  the same, waves that end at once (spin 0)", 1152, 256, 31232, 0);
    run<120>("  the same, 1024 wgs (one round of resident waves)", 1024, 256, 31232, 100);
    run<64>("  64 VGPRs", 1152, 256, 31232, 100);
    run<120>("  no LDS", 1152, 256, 0, 100);
    run<64>("  64 VGPRs, no LDS", 1152, 256, 0, 100);
    run<120>("  8-wave wgs (576)", 576, 512, 62464, 100);
    run<120>("  2-wave wgs (2304)", 2304, 128, 15616, 100);
    run<120>("  1-wave wgs (4608)", 4608, 64, 7808, 100);
    run<120>("  16-wave wgs (288)", 288, 1024, 124928, 100);
    run_big<0>("two 456-byte argument structs", 1024, 256, 31232, 100);
    run_big<3000>("  + 24 KB straight-line body", 1024, 256, 31232, 100);
    run_big<3000>("  + 24 KB straight-line body, 1152 wgs, spin 8 us", 1152, 256, 31232, 800);
    return 0;
}

// Vector-ALU issue rates on gfx950 for the instruction mix of the frame->FFT kernels: how many SIMD
// cycles a wave64 v_add/v_mul/v_fma costs in f32, packed f32 and f64, at 1, 2 and 4 waves per SIMD.
// Built and run by tools/ubench/run.sh on the GPU box; prints one table.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int REPS = 256;   // loop trips
constexpr int UNR = 16;     // independent instructions per trip

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long* cycles, double* sink, double seed)
{
    double d[UNR];
    float f[UNR];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[UNR];
#pragma unroll
    for (int i = 0; i < UNR; i++) {
        d[i] = seed + i + threadIdx.x;
        f[i] = (float)d[i];
        p[i] = f2{f[i], f[i] + 1.f};
    }
    double dm = 1.0000001 + seed * 1e-9, da = 1e-7 * seed;
    float fm = (float)dm, fa = (float)da;
    f2 pm = f2{fm, fm}, pa = f2{fa, fa};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fa));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fm));
            if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fm), "v"(fa));
            if (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pa));
            if (OP == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(da));
            if (OP == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
            if (OP == 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
            if (OP == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 10) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(fa));
            if (OP == 12) asm volatile("v_log_f32 %0, %1" : "=v"(f[i]) : "v"(fm));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double acc = 0;
#pragma unroll
    for (int i = 0; i < UNR; i++) acc += d[i] + f[i] + p[i].x + p[i].y;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (acc == 12345.678) sink[0] = acc;
}

typedef void (*kern_t)(unsigned long long*, double*, double);

int main()
{
    const char* names[] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32",
                           "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_mov_b32", "v_log_f32"};
    kern_t kerns[] = {k_rate<0>, k_rate<1>, k_rate<2>, k_rate<3>, k_rate<4>, k_rate<5>, k_rate<6>,
                      k_rate<7>, k_rate<8>, k_rate<9>, k_rate<10>, k_rate<11>, k_rate<12>};
    unsigned long long* dc;
    double* ds;
    CHECK(hipMalloc(&dc, 4096 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&ds, 8));
    printf("SIMD cycles per wave64 instruction (s_memtime ticks / instructions issued per SIMD), one workgroup on one CU\n");
    printf("%-16s %10s %10s %10s\n", "instruction", "1 wave/SIMD", "2 waves", "4 waves");
    for (int op = 0; op < 13; op++) {
        printf("%-16s", names[op]);
        for (int wps : {1, 2, 4}) {
            int threads = 256 * wps;   // 4 SIMDs x wps waves
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(kerns[op], dim3(1), dim3(threads), 0, 0, dc, ds, 1.0);
                CHECK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h(threads / 64);
            CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (auto v : h) mx = v > mx ? v : mx;
            // per SIMD: wps waves x REPS x UNR instructions in mx ticks
            printf(" %10.2f", (double)mx / ((double)wps * REPS * UNR));
        }
        printf("\n");
    }
    return 0;
}

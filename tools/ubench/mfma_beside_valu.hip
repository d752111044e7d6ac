// Does the matrix pipe take the mel reduction off the vector ALU's hands?  (north star: "MFMA only if the mel triangle apply is
// re-cast as a dense filter x freq GEMM and rocprof shows it wins"; mel/mel.go:122-152 is the loop in question.)
// The headline kernel is bound by vector-ALU ISSUE: a wave tile is ~775 float64 butterfly instructions plus
// an epilogue whose mel sums are 68 v_fma_f32 per lane (17 chunk steps x 4).  Re-cast as a banded [16 filters] x [4 bins] x
// [16 frame columns, 6 live] product the same sums are 55 v_mfma_f32_16x16x4_f32 per wave tile.  This program times, per wave
// and iteration, a block of 160 independent v_fma_f64 (the FFT's stand-in)
//   mode 0  alone,
//   mode 1  with 16 v_fma_f32 spread through it      (the vector form's share: 68 per 680),
//   mode 2  with 13 v_mfma_f32_16x16x4_f32 instead   (the matrix form's share: 55 per 680; four accumulators in rotation),
//   mode 3  13 MFMAs alone (the matrix pipe's own pace),
// at 1, 2 and 4 waves per SIMD, all CUs busy, and prints SIMD cycles per iteration and the cost of one added instruction.
// Built and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_beside_valu mfma_beside_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int REPS = 1024;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k_mix(unsigned long long* cycles, double* sink, double seed) {
    double d[16];
    float f[16];
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        d[i] = seed + i + threadIdx.x;
        f[i] = float(d[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const double dm = 1.0000001 + seed * 1e-9, da = 1e-7 * seed;
    const float fm = float(dm), fa = float(da);
    float wa = fm, pb = fa;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int blk = 0; blk < 10; blk++) {  // 10 x 16 = 160 float64 multiply-adds
            if (MODE != 3) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dm), "v"(da));
            }
            if (MODE == 1) {  // 16 float32 multiply-adds over the 10 blocks: 2, 1, 2, 1, ...
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[blk]) : "v"(fm), "v"(fa));
                if (blk < 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[10 + blk]) : "v"(fm), "v"(fa));
            }
            if (MODE == 2 || MODE == 3) {  // 13 MFMAs over the 10 blocks
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[blk & 3]) : "v"(wa), "v"(pb));
                if (blk < 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(blk + 2) & 3]) : "v"(wa), "v"(pb));
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    double a = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) a += d[i] + f[i];
#pragma unroll
    for (int i = 0; i < 4; i++) a += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (threadIdx.x % 64 == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a == 12345.678) sink[0] = a;
}

typedef void (*kern_t)(unsigned long long*, double*, double);

int main() {
    int dev = 0, cus = 256, clk = 0;
    CHECK(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
    unsigned long long* d_cyc;
    double* d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 16));
    CHECK(hipMalloc(&d_sink, 8));
    kern_t ks[4] = {k_mix<0>, k_mix<1>, k_mix<2>, k_mix<3>};
    const char* names[4] = {"160 v_fma_f64", "+ 16 v_fma_f32", "+ 13 v_mfma_f32_16x16x4_f32", "13 MFMAs alone"};
    printf("shader clock attribute %d kHz; %d CUs\n", clk, cus);
    printf("%-32s %14s %14s %14s\n", "per wave and iteration", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    double base[3] = {0, 0, 0};
    for (int m = 0; m < 4; m++) {
        double res[3];
        for (int wi = 0; wi < 3; wi++) {
            const int waves_per_simd = 1 << wi, threads = 64 * 4 * waves_per_simd;  // one workgroup per CU
            ks[m]<<<cus, threads>>>(d_cyc, d_sink, 1.0);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0));
            const int launches = 20;
            for (int l = 0; l < launches; l++) ks[m]<<<cus, threads>>>(d_cyc, d_sink, 1.0);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            res[wi] = double(ms) * 1e6 / launches / REPS;  // ns of wall time per loop iteration of the W resident waves of a SIMD
        }
        if (m == 0) for (int wi = 0; wi < 3; wi++) base[wi] = res[wi];
        printf("%-32s %11.1f ns %11.1f ns %11.1f ns", names[m], res[0], res[1], res[2]);
        if (m == 1 || m == 2) {
            const int n = m == 1 ? 16 : 13;
            printf("   SIMD time per added instruction: %.2f %.2f %.2f ns", (res[0] - base[0]) / n, (res[1] - base[1]) / (2 * n),
                   (res[2] - base[2]) / (4 * n));
        }
        printf("\n");
    }
    printf("(wall-clock ns per loop iteration, every resident wave running the same loop: at W waves per SIMD one iteration of all W\n"
           " waves takes that long, so the SIMD time one added instruction of one wave costs is (delta ns) / (W n); x 2.4 = cycles at 2.4 GHz)\n");
    return 0;
}

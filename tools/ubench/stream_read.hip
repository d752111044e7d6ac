// The measured HBM-read roof ("%roofline = achieved / measured_stream_read_GBps (a plain f32 read kernel on the same
// box)"): every lane streams float4 pieces of a buffer much larger than the 256 MB Infinity Cache and sums them; nothing
// is written but one float per workgroup whose condition never holds.  Measurement only -- not part of the product library:
// built by auditory_amd.build as tools/ubench/libstream_read.so and loaded by bench.py (roofline.measured_read_GBps) or run
// stand-alone (hipcc -DSTREAM_READ_MAIN).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;  // 16-byte loads a lane keeps in flight

__global__ __launch_bounds__(kThreads) void k_stream_read(const float4* __restrict__ src, size_t n16, float* sink) {
    const size_t stride = size_t(gridDim.x) * kThreads;
    size_t i = size_t(blockIdx.x) * kThreads + threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + (kUnroll - 1) * stride < n16; i += kUnroll * stride) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            s0 += v[u].x;
            s1 += v[u].y;
            s2 += v[u].z;
            s3 += v[u].w;
        }
    }
    for (; i < n16; i += stride) {
        const float4 v = src[i];
        s0 += v.x;
        s1 += v.y;
        s2 += v.z;
        s3 += v.w;
    }
    const float s = (s0 + s1) + (s2 + s3);
    if (s == 1.2345678e30f) sink[blockIdx.x] = s;  // never true for finite data: keeps the loads alive
}

}  // namespace

// Reads `bytes` (a multiple of 16) of `buf` `reps` times on `stream` between two HIP events; *gbps = bytes x reps / time.
// wgs_per_cu <= 0: 8.  Returns 0, or a hipError_t.
extern "C" int ubench_stream_read(const void* buf, size_t bytes, int reps, int wgs_per_cu, void* stream, float* sink, double* gbps,
                                  double* ms_per_pass) {
    if (!buf || !sink || bytes < 16 || reps < 1) return -1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int dev = 0, cus = 256;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return int(e);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = cus * (wgs_per_cu > 0 ? wgs_per_cu : 8);  // sink must hold `grid` floats (<= 256 x 32)
    hipEvent_t e0, e1;
    if ((e = hipEventCreate(&e0)) != hipSuccess) return int(e);
    if ((e = hipEventCreate(&e1)) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return int(e);
    }
    const size_t n16 = bytes / 16;
    k_stream_read<<<grid, kThreads, 0, st>>>(static_cast<const float4*>(buf), n16, sink);  // warm-up (TLB, clocks)
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) k_stream_read<<<grid, kThreads, 0, st>>>(static_cast<const float4*>(buf), n16, sink);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return int(e);
    if (ms_per_pass) *ms_per_pass = double(ms) / reps;
    if (gbps) *gbps = double(n16) * 16.0 * reps / (double(ms) * 1e-3) / 1e9;
    return 0;
}

#ifdef STREAM_READ_MAIN
int main() {
    const size_t bytes = size_t(2) << 30;
    void* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 256 * 32 * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, bytes);
    for (int w : {2, 4, 8, 16}) {
        double gbps = 0, ms = 0;
        const int rc = ubench_stream_read(buf, bytes, 10, w, nullptr, sink, &gbps, &ms);
        printf("stream read, 2 GiB x 10, %2d workgroups of 256 per CU: rc %d, %.3f ms per pass, %.0f GB/s\n", w, rc, ms, gbps);
    }
    return 0;
}
#endif

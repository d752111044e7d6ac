// The measured HBM-WRITE side of the whole-ProcessSegment kernel: what the chip sustains for
//   A  a plain streaming write (16 bytes per lane, whole cache lines): the write roof;
//   B1 PowerSegment-shaped stores as round 4 issued them: tensor [items][H = 201][T = 104] float32, one WAVE per tile of 6
//      consecutive steps, lane = (bin group, step): 4-byte stores, a bin's 6 steps = one 24-byte run, ~10 runs per instruction;
//   B2 the same runs as 12-byte stores, lane = (bin, half of the tile): 32 runs per instruction (round 5's wave_spectrum_halves);
//   B3 the four waves of a workgroup pooling their 4 consecutive tiles: 96-byte runs written as 16-byte stores, 6 lanes per
//      run (what a workgroup-cooperative transpose through LDS would issue);
// each with nothing but the stores in the kernel -- no FFT, no logarithm -- so the figure is the memory system's, not the
// vector ALU's.  Measurement only (not part of the product library).  hipcc -O2 --offload-arch=gfx950 stream_write.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {
constexpr int kH = 201, kT = 104, kTiles = 18;  // 104 steps = 17 tiles of 6 + one of 2

__global__ __launch_bounds__(256) void k_stream_write(float4* dst, size_t n16, float v) {
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) dst[i] = float4{v, v, v, v};
}

// the product's workgroup -> tile order (kernels.h tile_of_workgroup): the workgroups of one XCD (equal id mod 8) walk one
// contiguous run of tiles
__device__ __forceinline__ unsigned remap_wg(unsigned b, unsigned n, int remap) {
    if (!remap || n < 16) return b;
    const unsigned x = b & 7u, i = b >> 3, per = n >> 3, rem = n & 7u;
    return x * per + (x < rem ? x : rem) + i;
}

// wave = tile: item = tile / 18, t0 = 6 (tile % 18)
// skew_w / skew_g: the wave sleeps (tile % 4) x skew_w + (workgroup % 4) x skew_g periods of ~4 us before it stores -- the
// waves whose 24-byte runs share a cache line no longer write it at the same moment (what unequal progress through a real
// kernel's arithmetic does)
__global__ __launch_bounds__(256) void k_runs_b1(float* a, float* b, unsigned n_tiles, float v, int remap, int skew_w = 0, int skew_g = 0) {
    const unsigned wgx = remap_wg(blockIdx.x, gridDim.x, remap);
    const unsigned tile = wgx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    for (int s = 0; s < skew_w * int(tile & 3) + skew_g * int(wgx & 3); ++s) __builtin_amdgcn_s_sleep(127);
    const unsigned item = tile / kTiles, t0 = 6 * (tile % kTiles);
    const unsigned f = lane % 6, g = lane / 6;
    if (g >= 10 || t0 + f >= kT) return;
    size_t o = (size_t(item) * kH + g) * kT + t0 + f;
    for (unsigned k = g; k < kH; k += 10, o += 10 * kT) {
        a[o] = v;
        b[o] = v + 1.f;
    }
}
struct __attribute__((packed, aligned(4))) F3 {
    float x, y, z;
};
// pairing: 0 = the two halves of a run on neighbouring lanes (lane = 2 bin + half), 1 = on lanes 32 apart (lane = 32 half + bin)
__global__ __launch_bounds__(256) void k_runs_b2(float* a, float* b, unsigned n_tiles, float v, int remap, int pairing) {
    const unsigned tile = remap_wg(blockIdx.x, gridDim.x, remap) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const unsigned item = tile / kTiles, t0 = 6 * (tile % kTiles);
    if (t0 + 6 > kT) return;  // (the 2-step last tile: left out, 1/52 of the bytes)
    const unsigned h = pairing ? lane >> 5 : lane & 1, bb = pairing ? lane & 31 : lane >> 1;
    typedef unsigned u3 __attribute__((ext_vector_type(3)));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(a + size_t(item) * kH * kT, 0, kH * kT * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(b + size_t(item) * kH * kT, 0, kH * kT * 4, 0x00020000);
    const u3 va = {__float_as_uint(v), __float_as_uint(v), __float_as_uint(v)};
    for (unsigned i = 0; i < 7; ++i) {
        const int o = int(((32 * i + bb) * kT + t0 + 3 * h) * 4);
        __builtin_amdgcn_raw_buffer_store_b96(va, ra, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b96(va, rb, o, 0, 0);
    }
}
// a workgroup = 4 consecutive tiles of one item (24 steps = 96 bytes per bin row): wave w takes bins w, w + 4, ... ; lane =
// (row r of 10, 16-byte piece q of 6)
__global__ __launch_bounds__(256) void k_runs_b3(float* a, float* b, unsigned n_groups, float v, int remap) {
    const unsigned grp = remap_wg(blockIdx.x, gridDim.x, remap), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (grp >= n_groups) return;
    const unsigned item = grp / 4, t0 = 24 * (grp % 4);  // (4 groups of 24 steps = 96 of the 104: the tail left out)
    const unsigned q = lane % 6, r = lane / 6;
    if (r >= 10) return;
    for (unsigned k = 4 * r + w; k < kH; k += 40) {
        const size_t o = (size_t(item) * kH + k) * kT + t0 + 4 * q;
        *reinterpret_cast<float4*>(a + o) = float4{v, v, v, v};
        *reinterpret_cast<float4*>(b + o) = float4{v, v, v, v};
    }
}

// a workgroup = one item, whole rows: 416 bytes per bin = 26 lanes x 16 bytes, two rows per instruction; the item's tensor is
// contiguous, so this is a streaming write at item grain
__global__ __launch_bounds__(256) void k_runs_b4(float* a, float* b, unsigned n_items, float v, int remap) {
    const unsigned item = remap_wg(blockIdx.x, gridDim.x, remap), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (item >= n_items) return;
    const unsigned q = lane % 26, r = lane / 26;
    if (r >= 2) return;
    for (unsigned k = 2 * w + r; k < kH; k += 8) {
        const size_t o = (size_t(item) * kH + k) * kT + 4 * q;
        *reinterpret_cast<float4*>(a + o) = float4{v, v, v, v};
        *reinterpret_cast<float4*>(b + o) = float4{v, v, v, v};
    }
}
// a workgroup = half an item's steps (52 steps = 208-byte runs, 13 lanes x 16 bytes, four rows per instruction)
__global__ __launch_bounds__(256) void k_runs_b5(float* a, float* b, unsigned n_halves, float v, int remap) {
    const unsigned hf = remap_wg(blockIdx.x, gridDim.x, remap), w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (hf >= n_halves) return;
    const unsigned item = hf / 2, t0 = 52 * (hf % 2);
    const unsigned q = lane % 13, r = lane / 13;
    if (r >= 4) return;
    for (unsigned k = 4 * w + r; k < kH; k += 16) {
        const size_t o = (size_t(item) * kH + k) * kT + t0 + 4 * q;
        *reinterpret_cast<float4*>(a + o) = float4{v, v, v, v};
        *reinterpret_cast<float4*>(b + o) = float4{v, v, v, v};
    }
}

template <typename F>
double time_ms(F&& launch, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return double(ms) / reps;
}
}  // namespace

int main() {
    const unsigned items = 4096;
    const size_t tensor = size_t(items) * kH * kT;  // floats: 342.5 MB per tensor
    float *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, tensor * 4 + 64) != hipSuccess || hipMalloc(&b, tensor * 4 + 64) != hipSuccess) return 1;
    (void)hipMemset(a, 0, tensor * 4);
    (void)hipMemset(b, 0, tensor * 4);
    const double two = 2.0 * double(tensor) * 4.0;
    for (int wgs : {2, 4, 8, 16}) {
        const double ms = time_ms([&] {
            k_stream_write<<<256 * wgs, 256>>>(reinterpret_cast<float4*>(a), tensor / 4, 1.f);
            k_stream_write<<<256 * wgs, 256>>>(reinterpret_cast<float4*>(b), tensor / 4, 2.f);
        }, 10);
        printf("A  streaming write, 2 x 342 MB, %2d workgroups per CU: %.3f ms, %.0f GB/s\n", wgs, ms, two / (ms * 1e-3) / 1e9);
    }
    const unsigned n_tiles = items * kTiles;
    double ms = 0;
    for (int remap : {0, 1}) {
        ms = time_ms([&] { k_runs_b1<<<(n_tiles + 3) / 4, 256>>>(a, b, n_tiles, 3.f, remap); }, 10);
        printf("B1 24-byte runs as 4-byte stores (lane = bin group x step), one wave per 6-step tile, xcd remap %d: %.3f ms, %.0f GB/s\n",
               remap, ms, two / (ms * 1e-3) / 1e9);
    }
    for (int sw : {0, 1, 4})
        for (int sg : {0, 1, 4}) {
            if (sw == 0 && sg == 0) continue;
            ms = time_ms([&] { k_runs_b1<<<(n_tiles + 3) / 4, 256>>>(a, b, n_tiles, 3.f, 1, sw, sg); }, 10);
            printf("B1 xcd remap 1, waves skewed by (tile %% 4) x %d + (workgroup %% 4) x %d periods of ~4 us: %.3f ms, %.0f GB/s\n", sw, sg, ms,
                   two / (ms * 1e-3) / 1e9);
        }
    for (int remap : {0, 1})
        for (int pairing : {0, 1}) {
            ms = time_ms([&] { k_runs_b2<<<(n_tiles + 3) / 4, 256>>>(a, b, n_tiles, 4.f, remap, pairing); }, 10);
            printf("B2 24-byte runs as two 12-byte stores (lane = bin x half tile, halves %s), xcd remap %d: %.3f ms, %.0f GB/s\n",
                   pairing ? "32 lanes apart" : "on neighbouring lanes", remap, ms, two * (102.0 / 104.0) / (ms * 1e-3) / 1e9);
        }
    for (int remap : {0, 1}) {
        ms = time_ms([&] { k_runs_b3<<<items * 4, 256>>>(a, b, items * 4, 5.f, remap); }, 10);
        printf("B3 96-byte runs as 16-byte stores (a workgroup's four tiles pooled), xcd remap %d: %.3f ms, %.0f GB/s\n", remap, ms,
               two * (96.0 / 104.0) / (ms * 1e-3) / 1e9);
        ms = time_ms([&] { k_runs_b5<<<items * 2, 256>>>(a, b, items * 2, 6.f, remap); }, 10);
        printf("B5 208-byte runs (a workgroup = half an item's steps), xcd remap %d: %.3f ms, %.0f GB/s\n", remap, ms, two / (ms * 1e-3) / 1e9);
        ms = time_ms([&] { k_runs_b4<<<items, 256>>>(a, b, items, 7.f, remap); }, 10);
        printf("B4 whole 416-byte rows (a workgroup = an item), xcd remap %d: %.3f ms, %.0f GB/s\n", remap, ms, two / (ms * 1e-3) / 1e9);
    }
    return 0;
}

// TEST INFRASTRUCTURE ONLY -- a host stand-in for <hip/hip_runtime.h>.
//
// Lets the product's .hip sources be compiled UNCHANGED with g++ so that their indexing,
// barriers and arithmetic can be exercised on the CPU under AddressSanitizer / UBSan /
// ThreadSanitizer (GPU sanitizers are not available on this pool).  Every GPU thread is a real
// OS thread; __syncthreads() and the wave shuffles are real barriers, so a missing barrier is a
// data race TSan can see and a divergent barrier is a detected deadlock.
//
// This is NOT a CPU fallback of the product: libauditory_hip.so never contains or loads it;
// only tests/ builds tests/emul/libauditory_emul*.so and only `-m "not gpu"` tests load it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

struct alignas(16) double2 {
    double x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
struct alignas(8) float2 {
    float x, y;
};
struct alignas(8) uint2 {
    unsigned x, y;
};
struct alignas(16) uint4 {
    unsigned x, y, z, w;
};
inline void __builtin_amdgcn_sched_barrier(int) {}
// device_common.h absmax3 (one v_max3_f32 with abs modifiers on the GPU): NaN operands are ignored, as the instruction does
#define AUD_EMUL_ABSMAX3 1
inline float absmax3(float m, float x, float y) { return fmaxf(m, fmaxf(fabsf(x), fabsf(y))); }
inline float __builtin_amdgcn_logf(float v) { return log2f(v); }  // v_log_f32 is a base-2 logarithm
inline double __builtin_amdgcn_rcp(double v) { return double(float(1.0 / v)); }  // v_rcp_f64: an APPROXIMATION (~2^-26), refined by the caller
// same-value races the 20 x 10 wave kernel has by design (shadow lanes): ThreadSanitizer is told to look away
#if defined(__SANITIZE_THREAD__)
extern "C" void AnnotateIgnoreWritesBegin(const char*, int);
extern "C" void AnnotateIgnoreWritesEnd(const char*, int);
#define AUD_BENIGN_RACE_BEGIN() AnnotateIgnoreWritesBegin(__FILE__, __LINE__)
#define AUD_BENIGN_RACE_END() AnnotateIgnoreWritesEnd(__FILE__, __LINE__)
#endif
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_ACQ_REL); }  // callers pass wave-uniform values
inline unsigned __umulhi(unsigned a, unsigned b) { return unsigned((uint64_t(a) * uint64_t(b)) >> 32); }
inline int atomicMax(int* p, int v) {
    int cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (cur < v && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {}
    return cur;
}

// Buffer descriptors (raw buffers, stride 0): base pointer + size in bytes; a load whose byte offset -- taken as an
// UNSIGNED 32-bit number, so negative offsets are far out of range -- does not lie wholly inside [0, bytes) returns 0,
// as the hardware's range check does.  (The kernels never depend on what a load that straddles an end returns.)
struct __amdgpu_buffer_rsrc_t {
    const unsigned char* base;
    unsigned bytes;
};
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short /*stride*/, int num_records, int /*flags*/) {
    return __amdgpu_buffer_rsrc_t{static_cast<const unsigned char*>(p), unsigned(num_records)};
}
template <typename T>
inline T emul_buffer_load(const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset) {
    const uint64_t off = uint64_t(uint32_t(voffset)) + uint64_t(uint32_t(soffset));
    T v{};
    if (off + sizeof(T) <= uint64_t(r.bytes)) std::memcpy(&v, r.base + off, sizeof(T));
    return v;
}
struct emul_u32x2 {
    unsigned v[2];
    unsigned operator[](int i) const { return v[i]; }
};
inline emul_u32x2 __builtin_amdgcn_raw_buffer_load_b64(const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset, int) {
    return emul_buffer_load<emul_u32x2>(r, voffset, soffset);
}
struct emul_u32x4 {
    unsigned v[4];
    unsigned operator[](int i) const { return v[i]; }
};
inline emul_u32x4 __builtin_amdgcn_raw_buffer_load_b128(const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset, int) {
    return emul_buffer_load<emul_u32x4>(r, voffset, soffset);
}
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset, int) {
    return emul_buffer_load<unsigned>(r, voffset, soffset);
}
// stores: dropped when they do not lie wholly inside the descriptor's range, as the hardware drops them
inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset, int) {
    const uint64_t off = uint64_t(uint32_t(voffset)) + uint64_t(uint32_t(soffset));
    if (off + 4 <= uint64_t(r.bytes)) std::memcpy(const_cast<unsigned char*>(r.base) + off, &v, 4);
}
inline unsigned short __builtin_amdgcn_raw_buffer_load_b16(const __amdgpu_buffer_rsrc_t& r, int voffset, int soffset, int) {
    return emul_buffer_load<unsigned short>(r, voffset, soffset);
}


typedef int hipError_t;
typedef struct emul_stream* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }
// a small emulated device (4 CUs, 2 resident workgroups each) so that the persistent kernels' tile loops run more
// than one trip in the CPU tier
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return 0; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return 0; }

hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipGetLastError();
// events and inter-process handles: everything the emulator runs is synchronous and lives in one process, so an event is
// a token and a "handle" carries the pointer itself (the direct all-gather's indexing is what the CPU tier checks)
typedef struct emul_event* hipEvent_t;
enum { hipEventDisableTiming = 2, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t {
    char reserved[64];
};
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new char); return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<char*>(e); return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }  // (the emulator's launches and copies are synchronous)
enum { hipHostMallocDefault = 0 };
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
enum { hipHostRegisterPortable = 1, hipHostRegisterMapped = 2 };
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {
    *dev = host;
    return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
// (nothing is ever captured here: launches are synchronous)
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
inline hipError_t hipStreamGetCaptureInfo(hipStream_t, hipStreamCaptureStatus* st, unsigned long long* id) {
    if (st) *st = hipStreamCaptureStatusNone;
    if (id) *id = 0;
    return 0;
}
// the direct all-gather's arrival flags (capi_comm.hip): fine-grained memory is ordinary memory here, system-scope atomics are
// the compiler's, and the bounded poll's sleep is nothing
enum { hipDeviceMallocFinegrained = 1 };
inline hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
inline hipError_t hipMemset(void* dst, int value, size_t bytes) { std::memset(dst, value, bytes); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
inline void __builtin_amdgcn_s_sleep(int) {}
inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
// (the handle also carries the exporting process: "device" memory here is a process's own heap, so a handle of ANOTHER
// process cannot be mapped -- opening it fails the way a box without IPC support fails, instead of handing out a wild pointer)
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
    std::memset(h, 0, sizeof(*h));
    std::memcpy(h->reserved, &p, sizeof(p));
    const long long pid = (long long)getpid();
    std::memcpy(h->reserved + 8, &pid, sizeof(pid));
    return 0;
}
inline hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
    long long pid = 0;
    std::memcpy(&pid, h.reserved + 8, sizeof(pid));
    if (pid != (long long)getpid()) return 1;  // hipErrorInvalidValue
    std::memcpy(p, h.reserved, sizeof(*p));
    return 0;
}
inline hipError_t hipIpcCloseMemHandle(void*) { return 0; }
const char* hipGetErrorString(hipError_t e);

namespace aud_emul {
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
void sync_block();
// exchange one 8-byte payload between the lanes of the calling thread's wave
uint64_t wave_exchange(uint64_t mine, int src_lane);
const uint64_t* wave_publish(uint64_t mine);
void wave_release();
}  // namespace aud_emul

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    aud_emul::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { aud_emul::sync_block(); }
// wave-level ordering of LDS traffic (device_common.h wave_lds_fence): on the GPU a scheduling fence, here a
// real barrier over the wave's threads, so a missing one is a data race TSan reports
void aud_emul_wave_barrier();
inline void __builtin_amdgcn_wave_barrier() { aud_emul_wave_barrier(); }
#define __builtin_amdgcn_fence(order, ...) __atomic_thread_fence(order)

template <typename T>
inline T emul_shfl_any(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    raw = aud_emul::wave_exchange(raw, src_lane);
    T out;
    std::memcpy(&out, &raw, sizeof(T));
    return out;
}
inline int emul_lane() { return int(threadIdx.x & 63u); }
// the first lane's value for the whole wave (every lane of the wave calls it)
inline int __builtin_amdgcn_readfirstlane(int v) { return emul_shfl_any(v, emul_lane() & ~63); }
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
    const int lane = emul_lane();
    return emul_shfl_any(v, (lane & ~(width - 1)) | (src & (width - 1)));
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = emul_lane();
    const int src = lane ^ mask;
    return emul_shfl_any(v, (src & ~(width - 1)) == (lane & ~(width - 1)) ? src : lane);
}
template <typename T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    const int lane = emul_lane();
    const int src = lane + int(d);
    return emul_shfl_any(v, (src & ~(width - 1)) == (lane & ~(width - 1)) ? src : lane);
}
template <typename T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    const int lane = emul_lane();
    const int src = lane - int(d);
    return emul_shfl_any(v, (src >= 0 && (src & ~(width - 1)) == (lane & ~(width - 1))) ? src : lane);
}

// Data-parallel-primitive moves (the controls the kernels use): every lane publishes `src`, then takes the value of the
// lane the control names; lanes the row mask leaves out, and lanes whose source does not exist, keep `old`.
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int /*bank_mask*/, bool /*bound_ctrl*/) {
    const int l = emul_lane(), row = l >> 4, in_row = l & 15;
    uint64_t packed = uint32_t(src);
    const uint64_t* all = aud_emul::wave_publish(packed);
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);   // quad_perm
    else if (ctrl == 0x140) from = (l & ~15) | (15 - in_row);                          // row_mirror
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));                           // row_half_mirror
    else if (ctrl == 0x142) from = (row == 1 || row == 3) ? 16 * row - 1 : -1;         // row_bcast:15
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;                                 // row_bcast:31
    else std::abort();
    int out = old;
    if (((row_mask >> row) & 1) && from >= 0) out = int(uint32_t(all[from]));
    aud_emul::wave_release();
    return out;
}
inline int __builtin_amdgcn_readlane(int v, int lane) { return emul_shfl_any(v, lane); }
// mask of the wave's lanes whose predicate holds (every lane of the wave calls it)
inline uint64_t __builtin_amdgcn_ballot_w64(bool pred) {
    const uint64_t* all = aud_emul::wave_publish(pred ? 1u : 0u);
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) m |= (all[l] & 1u) << l;
    aud_emul::wave_release();
    return m;
}

// v_mfma_f32_16x16x4_f32 as the guide documents it: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// D register r of lane l = row 4 (l >> 4) + r, column l & 15; one k-ordered fmaf chain per element.
template <typename V>
inline V __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, V c, int, int, int) {
    const int l = emul_lane();
    const int col = l & 15, rq = l >> 4;
    uint32_t ab[2];
    std::memcpy(&ab[0], &a, 4);
    std::memcpy(&ab[1], &b, 4);
    uint64_t packed;
    std::memcpy(&packed, ab, 8);
    const uint64_t* all = aud_emul::wave_publish(packed);
    auto a_of = [&](int lane) { float f; std::memcpy(&f, reinterpret_cast<const char*>(&all[lane]), 4); return f; };
    auto b_of = [&](int lane) { float f; std::memcpy(&f, reinterpret_cast<const char*>(&all[lane]) + 4, 4); return f; };
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rq + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma(a_of(row + 16 * k), b_of(col + 16 * k), acc);
        c[r] = acc;
    }
    aud_emul::wave_release();
    return c;
}

inline float __int_as_float(int v) {
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}
inline unsigned __float_as_uint(float v) {
    unsigned u;
    std::memcpy(&u, &v, 4);
    return u;
}
inline float __uint_as_float(unsigned v) {
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}
inline int __float_as_int(float f) {
    int v;
    std::memcpy(&v, &f, 4);
    return v;
}

using std::max;
using std::min;

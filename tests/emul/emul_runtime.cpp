// TEST INFRASTRUCTURE ONLY -- runtime behind tests/emul/hip/hip_runtime.h (see its header).
// One OS thread per GPU thread of the running workgroup; workgroups run one after another.
#include <hip/hip_runtime.h>

#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#define EMUL_POISON(p, n) ASAN_POISON_MEMORY_REGION(p, n)
#define EMUL_UNPOISON(p, n) ASAN_UNPOISON_MEMORY_REGION(p, n)
#else
#define EMUL_POISON(p, n) ((void)0)
#define EMUL_UNPOISON(p, n) ((void)0)
#endif

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <cstdio>
#include <thread>
#include <vector>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace aud {
// the one dynamic-LDS allocation of the running workgroup (160 KiB per CU on MI355X)
alignas(16) unsigned char aud_dyn_lds[160 * 1024];
}  // namespace aud

namespace {

// Barrier whose participant count can shrink when threads leave the kernel early.  One 64-bit state word {live, waiting}
// changed by compare-and-swap -- exactly one thread per generation sees "everyone is here" and becomes the releaser -- and a
// generation counter the others sleep on through the futex syscall: a wake-up touches no mutex (the condition-variable form
// woke 255 threads into one lock per barrier, and the CPU tier spent 20 minutes of system time there).  The acquire / release
// pairs on `gen` are what orders the threads' memory accesses (and what ThreadSanitizer sees).
struct Barrier {
    std::atomic<uint64_t> state{0};  // live << 32 | waiting
    std::atomic<uint32_t> gen{0};
    static void futex(std::atomic<uint32_t>* addr, int op, uint32_t val) {
        syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), op, val, nullptr, nullptr, 0);
    }
    void reset(int n) { state.store(uint64_t(uint32_t(n)) << 32, std::memory_order_release); }
    void release() {
        gen.fetch_add(1, std::memory_order_acq_rel);
        futex(&gen, FUTEX_WAKE_PRIVATE, INT_MAX);
    }
    void arrive_and_wait() {
        const uint32_t g = gen.load(std::memory_order_acquire);
        uint64_t s = state.load(std::memory_order_relaxed);
        for (;;) {
            const uint32_t live = uint32_t(s >> 32), w = uint32_t(s) + 1;
            const bool last = w >= live;
            if (state.compare_exchange_weak(s, (uint64_t(live) << 32) | (last ? 0u : w), std::memory_order_acq_rel)) {
                if (last) return release();
                break;
            }
        }
        int spins = 0;
        while (gen.load(std::memory_order_acquire) == g) {
            if (++spins < 64) std::this_thread::yield();   // (a barrier of a few threads is usually passed within a yield or two)
            else futex(&gen, FUTEX_WAIT_PRIVATE, g);
        }
    }
    void drop() {
        uint64_t s = state.load(std::memory_order_relaxed);
        for (;;) {
            const uint32_t live = uint32_t(s >> 32) - 1, w = uint32_t(s);
            const bool fire = live > 0 && w >= live;
            if (state.compare_exchange_weak(s, (uint64_t(live) << 32) | (fire ? 0u : w), std::memory_order_acq_rel)) {
                if (fire) release();
                return;
            }
        }
    }
};

constexpr int kMaxThreads = 1024;
Barrier g_block;
Barrier g_wave[kMaxThreads / 64];
uint64_t g_xchg[kMaxThreads];
size_t g_lds_bytes = 0;
hipError_t g_last = hipSuccess;

}  // namespace

namespace aud_emul {

void sync_block() { g_block.arrive_and_wait(); }

uint64_t wave_exchange(uint64_t mine, int src_lane) {
    const int tid = int(threadIdx.x);
    const int w = tid >> 6;
    g_xchg[tid] = mine;
    g_wave[w].arrive_and_wait();
    const uint64_t got = g_xchg[(w << 6) | (src_lane & 63)];
    g_wave[w].arrive_and_wait();
    return got;
}

// every lane of the wave publishes one value; returns the wave's 64 slots for reading; wave_release() ends the
// read phase (two barriers per publish however many lanes are read)
const uint64_t* wave_publish(uint64_t mine) {
    const int tid = int(threadIdx.x);
    const int w = tid >> 6;
    g_xchg[tid] = mine;
    g_wave[w].arrive_and_wait();
    return &g_xchg[w << 6];
}

void wave_release() { g_wave[int(threadIdx.x) >> 6].arrive_and_wait(); }

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    const int nthr = int(block.x * block.y * block.z);
    if (nthr < 1 || nthr > kMaxThreads || lds_bytes > sizeof(aud::aud_dyn_lds)) {
        g_last = hipErrorInvalidValue;
        return;
    }
    g_lds_bytes = lds_bytes;
    const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
    // the workgroup's threads live for the whole launch and take the blocks one after another: between two blocks they meet
    // at `turn` (a barrier of their own that never shrinks), thread 0 prepares the next block -- LDS poison, the kernel-visible
    // barriers back to full strength -- and they meet again
    Barrier turn;
    turn.reset(nthr);
    auto prepare = [&]() {
        // poison LDS so that reads of never-written shared memory show up as NaNs / garbage
        EMUL_UNPOISON(aud::aud_dyn_lds, sizeof(aud::aud_dyn_lds));
        std::memset(aud::aud_dyn_lds, 0xFF, sizeof(aud::aud_dyn_lds));
        // LDS beyond what the launch asked for is off limits (ASan build traps the access)
        const size_t used = (lds_bytes + 7) & ~size_t(7);
        if (used < sizeof(aud::aud_dyn_lds))
            EMUL_POISON(aud::aud_dyn_lds + used, sizeof(aud::aud_dyn_lds) - used);
        g_block.reset(nthr);
        for (int w = 0; w * 64 < nthr; ++w) g_wave[w].reset(std::min(64, nthr - w * 64));
    };
    std::vector<std::thread> pool;
    pool.reserve(size_t(nthr));
    for (int t = 0; t < nthr; ++t) {
        pool.emplace_back([&, t]() {
            blockDim = block;
            gridDim = grid;
            threadIdx = dim3(unsigned(t) % block.x, (unsigned(t) / block.x) % block.y, unsigned(t) / (block.x * block.y));
            for (unsigned long b = 0; b < nblocks; ++b) {
                turn.arrive_and_wait();  // every thread has left the previous block (its drops included)
                if (t == 0) prepare();
                turn.arrive_and_wait();
                blockIdx = dim3(unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / ((unsigned long)grid.x * grid.y)));
                body();
                g_wave[t >> 6].drop();
                g_block.drop();
            }
        });
    }
    for (auto& th : pool) th.join();
    EMUL_UNPOISON(aud::aud_dyn_lds, sizeof(aud::aud_dyn_lds));
}

}  // namespace aud_emul

void aud_emul_wave_barrier() { g_wave[int(threadIdx.x) >> 6].arrive_and_wait(); }

hipError_t hipGetDeviceCount(int* n) {
    *n = 8;  // one emulated node
    return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    *p = std::malloc(bytes ? bytes : 1);  // exact size: ASan sees device-buffer overruns
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
    std::memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t) {
    std::memset(dst, value, bytes);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() {
    const hipError_t e = g_last;
    g_last = hipSuccess;
    return e;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulated HIP error"; }

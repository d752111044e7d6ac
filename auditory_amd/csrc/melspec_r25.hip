// Fast frame -> FFT -> power -> mel -> log kernel for win_samples = 400 (25 ms at 16 kHz, the
// parameter set the headline metric is named after).  400 has no power-of-two structure worth
// speaking of (2^4 * 5^2); the packed-real trick gives a 200-point complex FFT, done as 25 x 8:
//
//   * 8 lanes cooperate on a frame (8 frames per wave64, 16 frames per 128-thread workgroup);
//   * pass A: every lane pulls its 25 sample pairs z[8 n1 + j] straight from global memory
//     (8-byte loads, 64 B contiguous per frame and load) and runs a 25-point DFT (5 x 5) in
//     registers, then twiddles by W_200^(j k1);
//   * one LDS transpose, rows of 8 + pad complex (pitch 20 dwords = 4 * odd; frame pitch 528
//     dwords keeps the 16-byte row reads of neighbouring frames on different banks);
//   * pass B: the 25 rows of a frame are spread over its 8 lanes (rows j, j+8, j+16 and row 24
//     on lane 0): 8-point DFTs in registers, results written back in place, so the finished
//     spectrum Z[k1 + 25 k2] sits at row k1, column k2;
//   * real-FFT split: the 101 pairs (Z[k], Z[200-k]) of a frame are read back by its 8 lanes,
//     13 pairs each, giving the power bins k and 200-k;
//   * the power spectrum reuses the transpose buffer and the shared tile epilogue does the
//     optional power / log-power outputs and the mel reduction (8 filter groups x 16 frames).
//
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "device_common.h"

namespace aud {
namespace {

constexpr int kF = 16;    // frames per workgroup
constexpr int kNT = 128;  // threads per workgroup
constexpr int kM = 200;   // complex FFT length
constexpr int kN = 400;   // window length
constexpr int kH = 201;   // power bins
constexpr int kHp = 204;  // P row pitch: 4 * 51 elements

template <typename TT>
struct Layout {
    static constexpr int kRowC = (sizeof(TT) == 4) ? 10 : 9;        // 8 + pad complex per row
    static constexpr int kFrameC = (sizeof(TT) == 4) ? 264 : 226;   // 25 rows + pad
};

template <typename TT>
__global__ __launch_bounds__(128) void k_melspec_r25(const MelspecArgs a, const FastArgs e) {
    unsigned char* smem = dyn_lds();
    TT* Pbase = reinterpret_cast<TT*>(smem + e.p_off);          // power spectrum [16][kHp], over xch
    C2<TT>* xch = reinterpret_cast<C2<TT>*>(smem + e.xch_off);  // [16][25][kRowC]
    const int tid = threadIdx.x;
    const int f = tid >> 3;  // frame within the tile
    const int j = tid & 7;   // lane within the frame's 8-lane group
    const int T = a.T;

    const int tiles = (T + kF - 1) / kF;
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / tiles;
    const int t0 = (wg - item * tiles) * kF;
    const aud_item it = a.items[item];
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_400^k

    const SchedRegs sched = mel_schedule_fetch<kNT>(e, tid);  // issued ahead of the operand loads

    // ---- pass A operands: z[8 n1 + j] = (x[16 n1 + 2j], x[16 n1 + 2j + 1]), n1 = 0..24 ---------
    C2<TT> v[25];
    load_frame_pairs<TT, 25, 8, kN>(a, it, t0 + f, j, v);

    // the filter-group schedule and the chunked mel weights ride along into LDS behind the operand loads; first
    // used after the last barrier
    mel_schedule_store<kNT>(e, smem, tid, sched);
    stage_mel_weights<TT, kNT>(e, smem, tid);

    // ---- pass A: 25-point DFT over n1, twiddle W_200^(j k1) = W_400^(2 j k1), column write -------
    SmallDft<TT, 25>::run(v, tw, kN);
    {
        C2<TT>* col = xch + f * Layout<TT>::kFrameC + j;  // row k1, column n2 = j
        col[0] = v[0];
#pragma unroll
        for (int k1 = 1; k1 < 25; ++k1) col[k1 * Layout<TT>::kRowC] = cmul(v[k1], tw[2 * j * k1]);
    }
    __syncthreads();

    // the split's twiddles W_400^(j + 8 i): requested here so that pass B covers their latency (the compiler
    // cannot lift them over the barrier behind pass B by itself)
    C2<TT> wsp[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int k = j + 8 * i;
        wsp[i] = tw[k <= kM / 2 ? k : 0];
    }

    // ---- pass B: rows k1 = j, j+8, j+16 (and 24 on lane 0): 8-point DFT over n2, in place --------
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k1 = j + 8 * r;
        if (k1 < 25) {
            C2<TT>* row = xch + f * Layout<TT>::kFrameC + k1 * Layout<TT>::kRowC;
            C2<TT> u[8];
            if (sizeof(TT) == 4) {
                const C2x2<TT>* row2 = reinterpret_cast<const C2x2<TT>*>(row);
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) {
                    const C2x2<TT> pr = row2[n2];
                    u[2 * n2] = pr.a;
                    u[2 * n2 + 1] = pr.b;
                }
            } else {
#pragma unroll
                for (int n2 = 0; n2 < 8; ++n2) u[n2] = row[n2];
            }
            SmallDft<TT, 8>::run(u, nullptr, 0);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) row[k2] = u[k2];  // Z[k1 + 25 k2]
        }
    }
    __syncthreads();

    // ---- real-FFT split + power: pairs k = j + 8 i (k <= 100) and 200 - k --------------------------
    // X[k] = (E + T)/2, X[200-k] = conj(E - T)/2, E = Z[k] + conj Z[200-k], T = -i W_400^k (Z[k] - conj Z[200-k])
    TT plo[13], phi[13];
    {
        const C2<TT>* Z = xch + f * Layout<TT>::kFrameC;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int k = j + 8 * i;
            plo[i] = TT(0);
            phi[i] = TT(0);
            if (k <= kM / 2) {
                const int kb = (k == 0) ? 0 : kM - k;
                const C2<TT> A = Z[(k % 25) * Layout<TT>::kRowC + k / 25];
                const C2<TT> B = Z[(kb % 25) * Layout<TT>::kRowC + kb / 25];
                const C2<TT> w = wsp[i];
                const C2<TT> E = {A.x + B.x, A.y - B.y};
                const C2<TT> D = {A.x - B.x, A.y + B.y};
                const C2<TT> mD = {D.y, -D.x};
                const C2<TT> Tm = cmul(mD, w);
                const TT xr = E.x + Tm.x, xi = E.y + Tm.y;
                const TT yr = E.x - Tm.x, yi = E.y - Tm.y;
                plo[i] = TT(0.25) * (xr * xr + xi * xi);
                phi[i] = TT(0.25) * (yr * yr + yi * yi);
            }
        }
    }
    __syncthreads();  // every pair has been read: the power spectrum may now overwrite the buffer
    {
        TT* P = Pbase + f * kHp;
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int k = j + 8 * i;
            if (k <= kM / 2) {
                P[k] = plo[i];
                P[kM - k] = phi[i];  // k = 0 -> Nyquist bin 200; k = 100 -> the same bin, same value
            }
        }
        if (j < 3) P[kH + j] = TT(0);  // pad bins of the last 4-bin chunk
    }
    __syncthreads();

    tile_epilogue<TT, kNT, kF>(a, e, Pbase, kHp, smem, it, item, t0, tid);
}

}  // namespace

bool melspec_r25_supported(int N, int S, int compute_dtype, int n_chunks, int nf, FastArgs* out) {
    if (N != kN || S < 1 || nf < 1) return false;
    const int n_groups = kNT / kF, n_sched = n_groups + 1 + 4 * nf;
    const size_t sched = (size_t(n_sched) * 2 + 15) & ~size_t(15);  // kept as uint16 in LDS
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t framec = compute_dtype == AUD_F64 ? 226 : 264;
    const size_t xch = size_t(kF) * framec * 2 * tsz;
    const size_t pbytes = (size_t(kF) * kHp * tsz + 31) & ~size_t(31);
    const size_t w4 = (size_t(n_chunks) * 4 * tsz + 31) & ~size_t(31);
    size_t first = xch > pbytes ? xch : pbytes;
    first = (first + 31) & ~size_t(31);
    const size_t total = first + w4 + sched;
    if (total > 160 * 1024) return false;
    if (out) {
        out->sched_off = int(first + w4);
        out->n_sched = n_sched;
        out->n_groups = n_groups;
        out->direct = 1;
        out->xch_off = 0;
        out->p_off = 0;
        out->w4_off = int(first);
        out->lds_bytes = unsigned(total);
        out->n_chunks = n_chunks;
    }
    return true;
}

hipError_t melspec_r25_prepare(unsigned lds_bytes) {
    const void* fns[2] = {reinterpret_cast<const void*>(&k_melspec_r25<double>),
                          reinterpret_cast<const void*>(&k_melspec_r25<float>)};
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_melspec_r25(const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + kF - 1) / kF;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_melspec_r25<double>, grid, dim3(kNT), e.lds_bytes, st, a, e);
    else
        hipLaunchKernelGGL(k_melspec_r25<float>, grid, dim3(kNT), e.lds_bytes, st, a, e);
    return hipGetLastError();
}

}  // namespace aud

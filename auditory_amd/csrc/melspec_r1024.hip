// Fast frame -> FFT -> power -> mel -> log kernel for win_samples = 2048 (BASELINE config 5:
// 44.1 kHz, 46.44 ms windows, 128 mel).  The packed-real trick gives a 1024-point complex FFT,
// done as 16 x 16 x 4 with ONE WAVE PER FRAME (64 lanes x 16 complex points in registers):
//
//   n = 64 n1 + l,  l = 4 n2 + n3 = lane;   k = k1 + 16 (k2 + 16 k3)
//   stage 1  lane l: 16-point DFT over n1 of z[64 n1 + l] (operands straight from global memory,
//            512 B contiguous per load), twiddle W_1024^(l k1)
//   -- LDS transpose: rows k1 of 64 + 4 complex (row pitch 136 dwords: the stride-4 column reads
//            of stage 2 land on 8 disjoint bank windows) --
//   stage 2  lane (k1, n3): 16-point DFT over n2, twiddle W_64^(n3 k2)
//   stage 3  4-point DFT over n3 = across the 4 neighbouring lanes of a quad: two xor-shuffle
//            butterflies per value, no LDS (lane n3 ends up holding k3 = bitrev2(n3))
//   -- spectrum to LDS (256-blocks padded by 4 complex: conflict-free scatter), pairs (k, 1024-k)
//            read back with consecutive lanes on consecutive addresses --
//   split + power, power spectrum to LDS, shared tile epilogue (64 filter groups x 4 frames).
//
// 256 threads = 4 frames at a time; a workgroup walks 4 such groups (16 consecutive frames) so that
// the mel weights are copied to LDS once per 16 frames and the mel values leave in 64-byte runs
// (collected in an 8 KB LDS tile first).  Every LDS phase reuses the same 8.7 KB per frame.
// Reference semantics: sound/sndenv.go:438-478, dft/dft.go:53-85, mel/mel.go:120-153.
#include "device_common.h"

namespace aud {
namespace {

constexpr int kF = 4;      // frames in flight (one wave each)
constexpr int kSub = 4;    // groups of kF frames per workgroup
constexpr int kTile = kF * kSub;  // 16 frames per workgroup
constexpr int kNT = 256;
constexpr int kM = 1024;   // complex FFT length
constexpr int kN = 2048;   // window length
constexpr int kH = 1025;   // power bins
constexpr int kHp = 1028;  // P row pitch: 4 * 257 elements
constexpr int kRowC = 68;  // transpose row: 64 + 4 complex
constexpr int kZC = 1040;  // spectrum: 4 blocks of 256 + 4 pad complex
constexpr int kFrameC = 16 * kRowC;  // 1088 complex per frame >= kZC, >= kHp/2 (P as TT)

__device__ __forceinline__ int zpos(int k) { return k + 4 * (k >> 8); }

template <typename TT>
__global__ __launch_bounds__(256) void k_melspec_r1024(const MelspecArgs a, const FastArgs e) {
    unsigned char* smem = dyn_lds();
    TT* Pbase = reinterpret_cast<TT*>(smem + e.p_off);          // [4][kHp], reuses the frame regions
    C2<TT>* xch = reinterpret_cast<C2<TT>*>(smem + e.xch_off);  // [4][kFrameC]
    const int tid = threadIdx.x;
    const int f = tid >> 6;  // frame within the tile = wave
    const int l = tid & 63;  // lane
    const int T = a.T;

    const int tiles = (T + kTile - 1) / kTile;
    const int wg = int(tile_of_workgroup(blockIdx.x, gridDim.x, a.xcd_remap));
    const int item = wg / tiles;
    const int tile0 = (wg - item * tiles) * kTile;
    const aud_item it = a.items[item];
    const C2<TT>* __restrict__ tw = static_cast<const C2<TT>*>(a.tw);  // W_2048^k
    C2<TT>* fr = xch + f * kFrameC;  // this frame's LDS region
    float* melbuf = reinterpret_cast<float*>(smem + e.out_off);  // [nf][kTile]

    stage_mel_weights<TT, kNT>(e, smem, tid);

  for (int sub = 0; sub < kSub; ++sub) {
    const int t0 = tile0 + sub * kF;
    if (t0 >= T) break;  // uniform: nothing left in this item

    // ---- stage 1 operands: z[64 n1 + l] = (x[128 n1 + 2 l], x[128 n1 + 2 l + 1]) ------------------
    C2<TT> v[16];
    // (no int16 route here: it pushes this kernel from 162 to 194 VGPRs, i.e. from 3 waves per SIMD to 2)
    load_frame_pairs<TT, 16, 64, kN, false>(a, it, t0 + f, l, v);

    // ---- stage 1: DFT over n1, twiddle W_1024^(l k1) = W_2048^(2 l k1), row k1 column l -----------
    SmallDft<TT, 16>::run(v, nullptr, 0);
    fr[l] = v[0];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) fr[k1 * kRowC + l] = cmul(v[k1], tw[2 * l * k1]);
    __syncthreads();

    // ---- stage 2: lane (k1, n3): DFT over n2 of B[4 n2 + n3][k1], twiddle W_64^(n3 k2) = W_2048^(32 n3 k2)
    const int k1 = l >> 2, n3 = l & 3;
    {
        const C2<TT>* col = fr + k1 * kRowC + n3;
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) v[n2] = col[4 * n2];
    }
    SmallDft<TT, 16>::run(v, nullptr, 0);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) v[k2] = cmul(v[k2], tw[32 * n3 * k2]);

    // the split's twiddles W_2048^(l + 64 i): requested here so that stage 3 and the spectrum scatter cover their
    // latency (the compiler cannot lift them over the two barriers in between by itself)
    C2<TT> wsp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wsp[i] = tw[l + 64 * i];

    // ---- stage 3: 4-point DFT over n3 across the quad (lanes 4 k1 .. 4 k1 + 3) ----------------------
    // xor-2 butterfly, then xor-1 butterfly with the -i twiddle on the odd branch; lane n3 ends with
    // output index k3 = bitrev2(n3): n3 0,1,2,3 -> k3 0,2,1,3
    {
        const bool hi2 = (n3 & 2) != 0, hi1 = (n3 & 1) != 0;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            C2<TT> o;
            o.x = __shfl_xor(v[k2].x, 2, 64);
            o.y = __shfl_xor(v[k2].y, 2, 64);
            // lanes 0,1: d_n + d_{n+2};   lanes 2,3: d_{n-2} - d_n
            C2<TT> s = hi2 ? C2<TT>{o.x - v[k2].x, o.y - v[k2].y} : C2<TT>{v[k2].x + o.x, v[k2].y + o.y};
            // the odd input of the difference branch carries -i: lane 3 holds (d1 - d3) -> * (-i)
            if (hi2 && hi1) s = mul_mi(s);
            o.x = __shfl_xor(s.x, 1, 64);
            o.y = __shfl_xor(s.y, 1, 64);
            v[k2] = hi1 ? C2<TT>{o.x - s.x, o.y - s.y} : C2<TT>{s.x + o.x, s.y + o.y};
        }
    }
    __syncthreads();  // every stage-2 column has been read: the region may take the spectrum

    // ---- spectrum to LDS at its natural index k = k1 + 16 k2 + 256 k3 ---------------------------------
    {
        const int k3 = ((n3 & 1) << 1) | (n3 >> 1);
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) fr[zpos(k1 + 16 * k2 + 256 * k3)] = v[k2];
    }
    __syncthreads();

    // ---- real-FFT split + power: pairs k = l + 64 i (i = 0..7, k < 512) and 1024 - k; k = 512 on lane 0 --
    TT plo[8], phi[8], pmid = TT(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = l + 64 * i;
        const int kb = (k == 0) ? 0 : kM - k;
        const C2<TT> A = fr[zpos(k)];
        const C2<TT> B = fr[zpos(kb)];
        const C2<TT> w = wsp[i];  // W_2048^k
        const C2<TT> E = {A.x + B.x, A.y - B.y};
        const C2<TT> D = {A.x - B.x, A.y + B.y};
        const C2<TT> mD = {D.y, -D.x};
        const C2<TT> Tm = cmul(mD, w);
        const TT xr = E.x + Tm.x, xi = E.y + Tm.y;
        const TT yr = E.x - Tm.x, yi = E.y - Tm.y;
        plo[i] = TT(0.25) * (xr * xr + xi * xi);
        phi[i] = TT(0.25) * (yr * yr + yi * yi);
    }
    if (l == 0) {
        const C2<TT> z = fr[zpos(kM / 2)];  // X[512] = conj(Z[512])
        pmid = z.x * z.x + z.y * z.y;
    }
    __syncthreads();  // every pair has been read: the power spectrum may now overwrite the region
    {
        TT* P = Pbase + f * kHp;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = l + 64 * i;
            P[k] = plo[i];
            P[kM - k] = phi[i];  // k = 0 -> Nyquist bin 1024
        }
        if (l == 0) P[kM / 2] = pmid;
        if (l < 3) P[kH + l] = TT(0);  // pad bins of the last 4-bin chunk
    }
    __syncthreads();

    tile_epilogue<TT, kNT, kF, false>(a, e, Pbase, kHp, smem, it, item, t0, tid, melbuf, kTile, sub * kF);
    __syncthreads();  // the power spectrum is consumed: the next group may reuse the frame regions
  }

    // ---- the 16-frame mel tile leaves in 64-byte runs ---------------------------------------------------
    for (int w = tid; w < a.nf * kTile; w += kNT) {
        const int flt = w / kTile, c = w - flt * kTile;
        const int sstep = tile0 + c;
        if (sstep < T) a.mel[(size_t(item) * a.nf + flt) * T + sstep] = melbuf[flt * kTile + c];
    }
}

}  // namespace

bool melspec_r1024_supported(int N, int S, int compute_dtype, int n_chunks, int nf, FastArgs* out) {
    if (N != kN || S < 1 || nf < 1) return false;
    const size_t tsz = compute_dtype == AUD_F64 ? 8 : 4;
    const size_t frames = size_t(kF) * kFrameC * 2 * tsz;  // also covers P: 4 * 1028 * tsz
    const size_t w4 = (size_t(n_chunks) * 4 * tsz + 31) & ~size_t(31);
    const size_t first = (frames + 31) & ~size_t(31);
    const size_t outb = size_t(nf) * kTile * sizeof(float);  // the 16-frame mel tile
    // (no LDS copy of the filter-group schedule here: its 1.2 KB would cost the third workgroup per CU)
    const int n_groups = 256 / kF, n_sched = 0;
    const size_t total = first + w4 + outb;
    if (total > 160 * 1024) return false;
    if (out) {
        out->sched_off = 0;
        out->n_sched = n_sched;
        out->n_groups = n_groups;
        out->direct = 1;
        out->xch_off = 0;
        out->p_off = 0;
        out->w4_off = int(first);
        out->out_off = int(first + w4);
        out->lds_bytes = unsigned(total);
        out->n_chunks = n_chunks;
    }
    return true;
}

hipError_t melspec_r1024_prepare(unsigned lds_bytes) {
    const void* fns[2] = {reinterpret_cast<const void*>(&k_melspec_r1024<double>),
                          reinterpret_cast<const void*>(&k_melspec_r1024<float>)};
    for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_melspec_r1024(const MelspecArgs& a, const FastArgs& e, int compute_dtype, hipStream_t st) {
    const int tiles = (a.T + kTile - 1) / kTile;
    const dim3 grid(unsigned(a.n_items) * unsigned(tiles));
    if (compute_dtype == AUD_F64)
        hipLaunchKernelGGL(k_melspec_r1024<double>, grid, dim3(kNT), e.lds_bytes, st, a, e);
    else
        hipLaunchKernelGGL(k_melspec_r1024<float>, grid, dim3(kNT), e.lds_bytes, st, a, e);
    return hipGetLastError();
}

}  // namespace aud
